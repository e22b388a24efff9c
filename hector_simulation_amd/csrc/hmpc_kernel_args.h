// hmpc_kernel_args.h -- what the host side needs to know of the fused kernel (hmpc_kernel.h): its argument block, the status
// codes, the layout of the assembly debug dump.  Kept apart from the 2 900-line kernel template so that the host translation
// units (hmpc_capi.hip, hmpc_group.hip) compile in seconds and the kernel family builds in parallel (hmpc_variants.hip).
#pragma once
#include <stdint.h>

namespace hmpc {

struct KernelArgs {
  const unsigned char *records;
  int stride, batch, horizon;
  float dt, f_max;
  float *forces;     // [batch][12h]
  uint32_t *status;  // [batch]
  double *x64;       // optional [batch][12h]
  double *obj64;     // optional [batch]
  // assembly-only debug dump (hmpc_debug_assemble)
  int dbg_index;
  float *dbg_f;
  int *dbg_i;
  long long *prof;  // optional [batch][NPROF] per-phase shader-clock cycles (thread 0's view), profiling builds only
  const int *index_list;  // optional: workgroup b solves instance index_list[b] (re-solve of flagged instances)
  int warm;         // 1: block warm start of the working set (default), 0: cold start as the reference does
  // warm start across ticks (the reference cold-starts, SolverMPC.cpp:702): per instance the
  // final working set of the previous solve, one signed byte per ORIGINAL constraint row (8 nc h; +1 lower side,
  // -1 upper side, 0 inactive).  Read at the start (rows of step i are taken from saved step min(i + wset_shift, h-1)),
  // overwritten at the end.  nullptr = off.
  signed char *wset;
  int wset_shift;
  // Last-resort pass for instances stuck at a degenerate vertex (hmpc_resolve_failed): every bound is moved outward by
  // relax * (1 + frac(0.618 row)) -- a different amount per row, which separates the coinciding vertices.  0 = exact.
  double relax;
  // optional device counter: +1 for every instance this launch leaves flagged for the safe pass (working set full,
  // max-iter, infeasible, KKT); lets hmpc_download skip the status scan when nothing was flagged
  unsigned int *flagged;
  // Device-side safe pass (hmpc_set_device_repair): a fast launch appends the index of every instance it flags to
  // flag_list[0 .. flag_cap) through the per-launch counter flag_count; the safe launch that follows on the same stream
  // takes flag_list as its index_list and list_count = flag_count, so that workgroups beyond the count leave at once --
  // no host round trip between the two launches.
  int *flag_list;
  unsigned int *flag_count;
  int flag_cap;
  const unsigned int *list_count;
  // Parity hook (hmpc_debug_solve_external_qp): QP data handed in instead of assembled -- per instance the reduced Hessian
  // [ext_ld][ext_ld] and gradient [ext_ld] in the reference's reduced order (binary32 values, as the reference's own H_red /
  // g_red are widened floats; the upper triangle is read) and the per-step constraint block [8 NC][6 NC]
  // (SolverMPC.cpp:466-548 fmat).  The record still supplies the gait table (structure) and f_max; stages S, W, Q run
  // unchanged.  nullptr = off (the product path).
  const float *ext_H, *ext_g, *ext_Fc;
  int ext_ld;
  // cap on the active-set iterations, the analogue of the reference's nWSR = 500 (SolverMPC.cpp:706): 0 = the variant's own
  // bound.  Block rounds and switch passes count as one iteration each; the block start itself always completes (it is one
  // inversion that stands for ~20 single-row iterations; checking the cap inside it costs the 168-VGPR variant spills),
  // the cap is tested before every single-row iteration after it.  A solve that would need more ends as S_MAXITER.
  int iter_cap;
  // Size classes (device-resident batches whose widest reduced QP the host does not know): cls[inst] = stance leg-steps of
  // the instance, written by build_records_kernel / classify_records_kernel; a workgroup leaves at once unless
  // cls_lo <= cls[inst] <= cls_hi, so that every variant of the family is launched over the whole batch and each instance
  // is solved by the smallest one that holds it -- no host round trip.  nullptr = every workgroup runs.
  const unsigned char *cls;
  int cls_lo, cls_hi;
  // scratch for the variants that keep the packed Schur inverse in global memory (Smem::EGLOBAL): NMAX (NMAX + 1) / 2
  // doubles per WORKGROUP of the launch (indexed by blockIdx.x)
  double *e_scratch;
  // Hand-over of a solve whose working set outgrew the fast variant's on-chip capacity (round 6; replaces "flag S_WORKSET and
  // re-solve cold").  A fast 120-variable variant that finds a violated row while its working set is full writes its live
  // Goldfarb-Idnani state -- point x, x_u, multipliers u, the working set (act / slot / Wrow), the packed Schur inverse E and the
  // 6 x 6 register blocks of M = H^-1 -- to the instance's own slot of `spill` (slot = instance index; spill_stride bytes each,
  // spill_cap slots; layout: SpillLayout in hmpc_kernel.h) and leaves the slot number in spill_slot[inst] (-1: nothing saved).
  // The continuation -- the safe variant of the same shape, whose working set holds as many rows as there are variables,
  // launched over the flagged list with `resume` set -- re-assembles the instance's constraint data (cheap, bit-identical),
  // takes M, E and the state from the slot instead of inverting H and starting cold, goes on with the iteration where the
  // fast variant stopped, and marks the slot consumed.  nullptr / 0 = off (a full working set is flagged S_WORKSET as before and
  // re-solved cold).
  unsigned char *spill;
  size_t spill_stride;
  int spill_cap;
  int *spill_slot;  // [batch]
  // robot / contact constants (struct hmpc_params; defaults = the reference's literals): 1 / mass as the host's binary32 quotient
  // (what the compiler folds 1.0f / 9.0f to), body inertia diagonal, friction coefficient, toe / heel lever arms, gravity state
  float inv_mass, Ib[3], mu, lt, lh, gravity;
  // optional per-INSTANCE friction parameter (hmpc_set_instance_mu: terrain sweeps): mu_inst[inst] replaces mu; nullptr = off.  H does
  // not depend on it (only the friction rows of the constraint block do), so instances of one command-sweep group may differ in it
  const float *mu_inst;
  // command sweeps (MODE 1 kernels, hmpc_solve_command_sweep): groups of sweep_k consecutive records that differ in the reference
  // trajectory only; sweep_phase 0 = one workgroup per group forms M = H^-1 and writes it to sweep_m[group][36][NT], phase 1 =
  // one workgroup per instance solves with its group's M
  int sweep_k, sweep_phase;
  double *sweep_m;
  int resume;       // continuation launch: 1 = instances with a valid slot resume from it, the others start cold; 2 = ... the others are left alone
  int skip_ok;      // list launch: instances whose status word says ok (an earlier pass over the same list solved them) are left alone; 1: ok and ok-relaxed, 2: ok only
  // A reduced Hessian that is NOT positive definite (the binary32 assembly of the contract rounds H = 2 (B'SB + alpha) at ~6e-8 |H|;
  // with a 20-step horizon at 10x the nominal input ranges that exceeds its smallest eigenvalue ~2 alpha in 2 % of the instances).
  // The reference's qpOASES run answers a failed Cholesky factorisation of H (QProblem.cpp:2107-2121, QProblemB.cpp:1418-1431) by
  // regularising -- H += rho I, rho = |H|_F min(-pivot + eps, sqrt(eps)), eps = 1e3 * 2.221e-16 (QProblemB.cpp:1999-2031; Options
  // setToMPC: enableRegularisation, numRegularisationSteps = 1) --, solving that QP, and solving it once more with the gradient
  // g - rho x_1 (QProblem.cpp:1753-1860).  The safe variants do the same in three launches over the flagged instances:
  //   reg_step 0 (every safe launch): a sweep pivot <= 0 ends the instance as S_INDEFINITE, forces zeroed, the pivot in reg_rho[inst];
  //   reg_step 1: instances whose status is S_INDEFINITE: rho from |H|_F and the pivot, left in reg_rho[inst]; H + rho I; solved:
  //               x_1 in the force buffer, status S_REG_STEP;
  //   reg_step 2: instances whose status is S_REG_STEP: H + rho I, g - rho x_1; the answer, status S_OK.
  int reg_step;
  // device-side chain: the safe launch (reg_step 0) appends the instances it ends as S_INDEFINITE to this list, which the two
  // regularisation launches then run over (a short list of its own: their grids stay small).  nullptr = off (host-driven repair
  // builds its list from the status words)
  int *reg_list;
  unsigned int *reg_count;
  int reg_cap;
  double *reg_rho;  // [batch]: the pivot (after reg_step 0), then rho; set for every launch over an index list
};
constexpr int NPROF = 32;
enum : int { P_ASM = 0, P_HG, P_SWEEP, P_XU, P_SEL, P_D, P_ED, P_W, P_MV, P_T1, P_UPD, P_POLISH, P_FINAL, P_TOTAL, P_BLOCK, P_B_S0, P_B_INV, P_B_DROP, P_SEL_A, P_A0, P_A1, P_A2, P_G };

// offsets (in floats) of the debug dump, shared with the host
template <int NMAX, int NC = 2>
struct DbgLayout {
  static constexpr int H = 0, G = NMAX * NMAX, FC = G + NMAX, LB = FC + 48 * NC * NC, UB = LB + 8 * NC * 20,
                       X0 = UB + 8 * NC * 20, ACD = X0 + 16, BCD = ACD + 176, TOTAL = BCD + 80 * NC;
};

enum : int { S_OK = 0, S_MAXITER = 1, S_INFEASIBLE = 2, S_TOO_LARGE = 3, S_KKT = 4, S_WORKSET = 5, S_OK_RELAXED = 6, S_SWEEP_MISMATCH = 7, S_INDEFINITE = 8, S_REG_STEP = 9 };

}  // namespace hmpc
