// hmpc_variants.hip -- instantiates the kernel family of hmpc_kernel.h.  Built once per group (-DHMPC_VARIANT_GROUP=k,
// hector_simulation_amd/build.py) into separate objects: the 2 900-line kernel template costs 5-20 s per instantiation, and
// the twelve variants (x solve / assemble-only) compile side by side instead of in one 60-second translation unit.
#include <hip/hip_runtime.h>

#include "hmpc_kernel.h"
#include "hmpc_variants.h"

#ifndef HMPC_VARIANT_GROUP
#error "compile with -DHMPC_VARIANT_GROUP=0..3 (hector_simulation_amd/build.py does)"
#endif

namespace {
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT, int MODE>
kernel_fn assemble_kernel() {
  if constexpr (MODE == 0) return hmpc::hmpc_kernel<NMAX, HMAX, NT, QCAP, true, NC, BPT, 0>;
  else return nullptr;
}
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT, int MODE>
Variant make_variant() {
  static_assert(sizeof(hmpc::Smem<NMAX, HMAX, NT, QCAP, NC, BPT>) <= 160 * 1024, "LDS budget of a gfx950 CU");
  static_assert(BPT == 1 || NT >= 512 || sizeof(hmpc::Smem<NMAX, HMAX, NT, QCAP, NC, BPT>) <= 80 * 1024, "two workgroups per CU");
  using SM = hmpc::Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  // (the same shape test as SHAPE_HANDOVER / SPILLS / RESUMABLE in hmpc_kernel.h)
  constexpr bool handover = NMAX == 120 && NT == 256 && NC == 2 && BPT == 1 && QCAP != 0;
  static_assert(!handover || QCAP == HMPC_QCAP_FAST || QCAP >= HMPC_QCAP_CONT, "hand-over: the fast variants save, capacities from HMPC_QCAP_CONT on resume");
  return Variant{NMAX, HMAX, NT, QCAP, NC, MODE, hmpc::hmpc_kernel<NMAX, HMAX, NT, QCAP, false, NC, BPT, MODE>,
                 // (the assembly-only debug kernel belongs to the MODE 0 variant of the shape: built there, never here)
                 assemble_kernel<NMAX, HMAX, NT, QCAP, NC, BPT, MODE>(), sizeof(SM),
                 hmpc::DbgLayout<NMAX, NC>::TOTAL,
                 (handover && MODE == 0 && QCAP < HMPC_QCAP_CONT) ? hmpc::SpillLayout<SM, NT, BPT>::stride_for(QCAP) : 0, handover && QCAP >= HMPC_QCAP_CONT && QCAP < NMAX};
}
}  // namespace

#define HMPC_DEFINE_VARIANT(IDX, GRP, NMAX, HMAX, NT, QCAP, NC, BPT, MODE) HMPC_DEFINE_VARIANT_##GRP(IDX, NMAX, HMAX, NT, QCAP, NC, BPT, MODE)
#define HMPC_DEFINE_IT(IDX, NMAX, HMAX, NT, QCAP, NC, BPT, MODE) \
  Variant hmpc_variant_##IDX() { return make_variant<NMAX, HMAX, NT, QCAP, NC, BPT, MODE>(); }
#define HMPC_SKIP_IT(IDX, NMAX, HMAX, NT, QCAP, NC, BPT, MODE)
#if HMPC_VARIANT_GROUP == 0
#define HMPC_DEFINE_VARIANT_0 HMPC_DEFINE_IT
#else
#define HMPC_DEFINE_VARIANT_0 HMPC_SKIP_IT
#endif
#if HMPC_VARIANT_GROUP == 1
#define HMPC_DEFINE_VARIANT_1 HMPC_DEFINE_IT
#else
#define HMPC_DEFINE_VARIANT_1 HMPC_SKIP_IT
#endif
#if HMPC_VARIANT_GROUP == 2
#define HMPC_DEFINE_VARIANT_2 HMPC_DEFINE_IT
#else
#define HMPC_DEFINE_VARIANT_2 HMPC_SKIP_IT
#endif
#if HMPC_VARIANT_GROUP == 3
#define HMPC_DEFINE_VARIANT_3 HMPC_DEFINE_IT
#else
#define HMPC_DEFINE_VARIANT_3 HMPC_SKIP_IT
#endif
HMPC_VARIANT_TABLE(HMPC_DEFINE_VARIANT)
