// hmpc_variants.h -- the table of kernel variants, shared between the host side (hmpc_capi.hip: picks and launches) and
// hmpc_variants.hip (instantiates; compiled once per group -DHMPC_VARIANT_GROUP=0..3 so that the groups build in parallel).
#pragma once
#include <stddef.h>

#include "hmpc_kernel_args.h"

#ifndef HMPC_QCAP_FAST
#define HMPC_QCAP_FAST 64  // working-set capacity of the fast 120-variable h <= 10 variant (49 KB LDS: three per CU)
#endif
#ifndef HMPC_QCAP_CONT
#define HMPC_QCAP_CONT 96  // ... of the CONTINUATION variant of the 120-variable shapes (70 KB LDS: two per CU): takes over the solves whose working set outgrew the fast variant's
#endif
// (not switches: each is pinned from both sides by the LDS budget and the staging areas that alias the solver state --
//  static_asserts in hmpc_kernel.h / hmpc_variants.hip -- 152 rows are what 160 KB leave next to the wide variant's mat-vec staging,
//  96 what 80 KB, two workgroups per CU, leave the three-contact one)
constexpr int HMPC_QCAP_WIDE = 152;  // ... of the 240-variable variant (double support over h = 11 .. 20)
constexpr int HMPC_QCAP_3C = 96;     // ... of the fast three-contact variant (256 threads, two register blocks each, <= 80 KB LDS: two per CU)

typedef void (*kernel_fn)(hmpc::KernelArgs);

struct Variant {
  int nmax, hmax, nt, qcap, nc;
  int mode;  // 0: a workgroup per instance; 1: command sweeps (a workgroup per chunk of instances sharing state and gait)
  kernel_fn solve, assemble;
  size_t smem;
  int dbg_floats;
  // hand-over of a full working set (KernelArgs::spill): bytes of one slot when this variant SAVES its state (fast 120-variable
  // variants), 0 otherwise; resumes = this variant can continue from such a slot (the safe variants of the same shape)
  size_t spill_stride;
  bool resumes;
};

// index = position in hmpc_capi.hip's variants(); (NMAX, HMAX, NT, QCAP, NC, BPT, MODE), group = translation unit that builds it.
// MODE 1 = command sweeps (a workgroup solves a chunk of instances that share state and gait on one inverse, hmpc_kernel.h).
// The groups are balanced by compile time (the two-blocks-per-thread and 512-thread variants are the slow ones).
#define HMPC_VARIANT_TABLE(X)                             \
  X(0, 0, 60, 10, 128, 60, 2, 1, 0)                       \
  X(1, 0, 120, 10, 256, HMPC_QCAP_FAST, 2, 1, 0)          \
  X(2, 0, 60, 20, 128, 60, 2, 1, 0)                       \
  X(3, 1, 120, 20, 256, HMPC_QCAP_FAST, 2, 1, 0)          \
  X(4, 1, 120, 10, 256, 120, 2, 1, 0)                     \
  X(5, 1, 120, 20, 256, 120, 2, 1, 0)                     \
  X(6, 2, 180, 10, 256, HMPC_QCAP_3C, 3, 2, 0)            \
  X(7, 3, 180, 10, 512, 140, 3, 1, 0)                     \
  X(8, 2, 180, 10, 512, 100, 3, 1, 0)                     \
  X(9, 3, 240, 20, 512, HMPC_QCAP_WIDE, 2, 2, 0)          \
  X(10, 3, 240, 20, 512, 0, 2, 2, 0)                      \
  X(11, 2, 180, 10, 512, 0, 3, 1, 0)                      \
  X(12, 0, 120, 10, 256, HMPC_QCAP_CONT, 2, 1, 0)         \
  X(13, 1, 120, 20, 256, HMPC_QCAP_CONT, 2, 1, 0)         \
  X(14, 2, 120, 10, 256, HMPC_QCAP_FAST, 2, 1, 1)         \
  X(15, 3, 60, 10, 128, 60, 2, 1, 1)
constexpr int HMPC_VARIANT_GROUPS = 4;

#define HMPC_DECLARE_VARIANT(IDX, GRP, NMAX, HMAX, NT, QCAP, NC, BPT, MODE) Variant hmpc_variant_##IDX();
HMPC_VARIANT_TABLE(HMPC_DECLARE_VARIANT)
#undef HMPC_DECLARE_VARIANT
