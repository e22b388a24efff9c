// hmpc_kernel.h -- the fused assembly + QP-solve kernel (one workgroup per MPC instance, gfx950).
//
// Replaces, for a whole batch at once, the reference's per-tick CPU path
//   update_problem_data -> solve_mpc -> qpOASES::QProblem::init
//   (ConvexMPC/convexMPC_interface.cpp:83-103, ConvexMPC/SolverMPC.cpp:371-738, third_party/qpOASES/src/QProblem.cpp:316).
//
// Compiled for three waves per SIMD where the LDS footprint allows it (168 VGPRs; see WAVES_PER_EU_* below).
// Phases (all state lives in LDS / registers; HBM sees only the ~716 B record in and 12h floats + 1 word out):
//   A  assembly in binary32 under the HMPC-A1 arithmetic contract (bit-identical to the CPU oracle):
//      trig -> scalar algebra -> Acd^k, Phi_k = Acd^k Bcd -> tracking error -> swing elimination tables
//      -> H = 2(B'SB + alpha) on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 = k-ordered fmaf chain), g.
//      B_qp is block lower-triangular Toeplitz, so all blocks of one block-diagonal of H are prefixes of ONE chain: the
//      120/180-variable h <= 10 variants run h chains and read a finished block off after every step (bit-identical).
//   S  M = H^-1 in binary64 by n symmetric sweeps.  The matrix lives in REGISTERS for the rest of the kernel: the
//      reduced variables are ordered leg-step by leg-step ([F(3), M(3)] per stance leg-step), and thread t owns the
//      6x6 block M(e,e') between two leg-steps (e <= e'): 210 blocks for 20 leg-steps -- or BPT = 2 such blocks (template
//      parameter): the three-contact variant (465 blocks on 256 threads, two workgroups per CU) and the wide one (820
//      blocks on 512 threads: double support over h = 11..20).
//   W  block warm start, in rounds: every moment / line-contact row (rows 4-6 of a leg-step's 8) and one friction row per
//      axis (rows 0-3) violated at the unconstrained minimiser enter the working set at once.  Their Schur matrix N M N'
//      is formed block-locally (a row touches one leg-step, so n_i' M n_j needs only the 6x6 block its owner already
//      holds), inverted -- since round 5 by the same 4 x 4 block pivots on v_mfma_f64_16x16x4_f64 as stage S (schur_invert: 3 x 3
//      / 4 x 4 / 5 x 5 tiles for the 120-variable / three-contact / wide variants; the 128-thread and the safe variants keep the
//      register-resident two-pivots-per-barrier sweeps) --, rows of the foot-x moment window whose multiplier comes out negative are switched
//      to their other bound, other rows with a negative multiplier are removed again -> a valid Goldfarb-Idnani state,
//      ~20 iterations saved.  While enough further rows are violated at the point reached, another round takes the
//      working set plus all of them.
//   Q  dual active set (Goldfarb-Idnani, range-space form) from that state: Schur inverse E = (N M N')^-1 kept
//      explicitly (bordering / Schur-complement downdates: no triangular solves), M applied in place from the register
//      blocks with a fixed-order staged reduction (deterministic).  One refinement step of the multipliers at the end (HMPC_REFINE).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#ifndef HMPC_QCAP_CONT
#define HMPC_QCAP_CONT 96  // working-set capacity of the continuation variant of the 120-variable shapes (70 KB of LDS: two workgroups per CU)
#endif
#ifndef HMPC_REFINE
#define HMPC_REFINE 1  // corrections u += E (b_W - N_W x(u)) applied to the multipliers of the final working set
#endif

#include "hmpc_math.h"

#include "hmpc_kernel_args.h"

namespace hmpc {
// per-phase shader-clock profile of developer builds (-DHMPC_PROFILE, scripts/phase_profile.py); an empty object otherwise
#ifdef HMPC_PROFILE
struct Prof {
  long long pt, pt0, acc[NPROF];
  __device__ __forceinline__ Prof() : pt(clock64()), pt0(pt) {
    for (int i = 0; i < NPROF; ++i) acc[i] = 0;
  }
  __device__ __forceinline__ void mark(int ph) {
    const long long n = clock64();
    acc[ph] += n - pt;
    pt = n;
  }
  __device__ __forceinline__ void flush(const KernelArgs &args, int inst) {
    if (threadIdx.x == 0 && args.prof) {
      acc[P_TOTAL] = clock64() - pt0;
      for (int i = 0; i < NPROF; ++i) args.prof[(size_t)inst * NPROF + i] = acc[i];
    }
  }
};
#define PROF_MARK(ph) prof.mark(ph)
#define PROF_FLUSH() prof.flush(args, inst)
#else
struct Prof {};
#define PROF_MARK(ph)
#define PROF_FLUSH()
#endif
#define PROF_DECL Prof prof
}  // namespace hmpc

namespace hmpc {

// record field offsets in floats (hector_simulation_amd/records.py).  NC = 2 is the reference's update_data_t; NC = 3 is
// the extension record with a hand contact (its frame Rhand and force cap travel in the record).
template <int NC>
struct RecLayout {
  static constexpr int P = 0, V = 3, Q = 6, W = 10, R = 13, JA = R + 3 * NC, YAW = JA + 10, WT = YAW + 1, AL = WT + 12,
                       RH = AL + 6 * NC, FMH = RH + 9, NF = (NC == 2) ? RH : FMH + 1;
};


constexpr int GS = 6;  // variables per stance leg-step: force (3) then moment (3)


// BPT = register blocks per thread (1: one 6x6 block each, NT >= NG(NG+1)/2; 2: the three-contact variant on 256 threads,
// 465 blocks, two workgroups per CU -- which also needs its LDS under 80 KB: the staging of H is filled and drained in two
// passes over the block-diagonals and the staged mat-vec partials in two halves of the source leg-steps).
template <int NMAX, int HMAX, int NT, int QCAP, int NC = 2, int BPT = 1>
struct Smem {
  static constexpr int U = 6 * NC;       // variables per horizon step before elimination: F of each contact, then M
  static constexpr int PS = 13 * U;      // floats per Phi_k
  static constexpr int NG = NMAX / GS;   // leg-steps (blocks per matrix side)
  static constexpr int MMAX = NG * 8;    // constraint rows
  static constexpr int NW = NT / 64;
  // working-set capacity (packed Schur inverse E); QCAP = NMAX can never overflow.  QCAP = 0: capacity NMAX with E in GLOBAL
  // memory (a per-workgroup scratch slice, L2-resident) instead of LDS -- the safe pass of the wide variant, whose 240 x 240
  // packed triangle (231 KB) no CU's LDS holds
  static constexpr bool EGLOBAL = (QCAP == 0);
  static constexpr int QMAX = EGLOBAL ? NMAX : QCAP;
  static constexpr int EP_LDS = EGLOBAL ? 1 : QMAX * (QMAX + 1) / 2;
  static constexpr int RECW = ((RecLayout<NC>::NF + 12 * HMAX) * 4 + NC * HMAX + 15) / 16 * 4;  // record words
  static constexpr bool FULLBLK = (HMAX <= 10 && NMAX >= 120);  // staging layout of H (struct Asm)
  // FULLBLK staging passes: pass p holds the blocks of the block-diagonals d in [hs_dlo(p), hs_dlo(p+1))
  static constexpr int HSP = (NC == 3 && BPT == 2) ? 2 : 1;
  static constexpr int HS_D0 = 3;  // h = 10: 27 blocks on d < 3, 28 on d >= 3
  static constexpr int hs_dlo(int p) { return p <= 0 ? 0 : (p >= HSP ? HMAX : HS_D0); }
  static constexpr int hs_off(int d, int h) { return d * h - d * (d - 1) / 2; }  // blocks on the diagonals before d
  static constexpr int HS_BLOCKS = (HSP == 1) ? HMAX * (HMAX + 1) / 2
                                              : (hs_off(HS_D0, HMAX) > HMAX * (HMAX + 1) / 2 - hs_off(HS_D0, HMAX)
                                                     ? hs_off(HS_D0, HMAX) : HMAX * (HMAX + 1) / 2 - hs_off(HS_D0, HMAX));
  // staged mat-vec: partials of STH halves of the source leg-steps at a time (ST has NG / STH rows)
  static constexpr int STH = (BPT == 2) ? 2 : 1;
  static constexpr int STR = NG / STH;
  static_assert(NG % 2 == 0 && (STH == 1 || STH == 2), "the staged mat-vec sums two groups of NG/2 source leg-steps");

  double g[NMAX];        // gradient, sweep order
  double Cn[NC][8][6];   // per-contact constraint normals (columns: F then M of that contact)
  double ub7[NG];        // Fz cap f_max*gait of each stance leg-step
  double relax;          // args.relax (see row_lo in the kernel)
  double sc7[NG];        // ... and the unit scale of that row: 1/ub7 where ub7 > 1 (read where used: not a live register pair)
  unsigned char vstep[NMAX], vcomp[NMAX];  // reference-order reduced variable -> horizon step, component (0..11)
  unsigned char o2s[NMAX], s2o[NMAX];      // reference order <-> sweep order (leg-step major)
  unsigned short sinfo[NMAX];              // sweep-order variable -> horizon step | component << 8 (loader of the matrix-core sweeps)
  // matrix-core sweeps: power-of-two diagonal scaling, H~ = 2^k H 2^k with k_i = -floor(log2(H_ii) / 2) -- exact in binary
  // floating point in both directions (H_ii spans 2e-4 .. 500; the 4 x 4 pivot blocks of the scaled matrix are far better
  // conditioned than the raw ones, which is what the explicitly inverted pivot block needs)
  static constexpr bool MFS2 = (NMAX == 120 && NT == 256 && BPT == 1 && NC == 2);  // shapes whose fast variants sweep on the matrix cores
  static constexpr bool MFS3 = (NMAX == 180 && NT == 256 && BPT == 2 && NC == 3) ||
                               (NMAX == 240 && NT == 512 && BPT == 2 && NC == 2 && QCAP != 0);  // (round 5: the wide variant, 15 x 15 tiles on eight waves)
  signed char kexp[(MFS2 || MFS3) ? 16 * ((NMAX + 15) / 16) : 1];
  unsigned char rmap[U * HMAX];            // original variable U*step+comp -> sweep index (255 = eliminated)
  unsigned char ls_leg[NG], ls_step[NG];
  int n, m, nls, pad0;

  struct Asm {
    uint32_t rec[RECW];
    float sc[10][2];
    float sc234[2][2];
    float rpy[3];
    float ypsc[4];  // cy, sy, cp, sp
    float Acd[169], Bcd[PS], x0[13], W[13], Fc[8 * NC * U];
    float Apow[2 * 169];
    float Phi[HMAX * PS];  // Phi_k = Acd^k Bcd; S Phi_k = fl(w_s Phi_k) is formed where it is consumed (one multiply)
    float e[13 * HMAX];
    unsigned char pre_nl[HMAX], pre_nv[HMAX], pre_st[HMAX];  // per step: leg-steps / variables before it, stance bits
    // H in binary32 (exact), staged between the matrix-core phase and the register blocks of the sweeps.
    //   FULLBLK (120/180 variables at h <= 10): every U x U block (a <= b) of the UNREDUCED matrix, row-major, ordered by
    //     block-diagonal: block (a, a+d) at hs_off(d) + a -- written by the Toeplitz chains with no index arithmetic
    //     (entries of swing leg-steps are simply never read); + one spare word per lane for the padding lanes of the tiles;
    //     with HSP = 2 the area holds the diagonals of one pass at a time;
    //   otherwise: upper triangle over the reduced variables in reference order, rows i and NMAX-1-i folded into one.
    float Hs[FULLBLK ? HS_BLOCKS * U * U + 64 : (NMAX / 2) * (NMAX + 1)];
  };
  struct Rec {
    double val, raw, cn[6];
    int idx, side, pad1, pad2;
  };
  struct Sol {
    alignas(16) double x[NMAX], xu[NMAX], z[NMAX], w[NMAX];
    // staged partials of the in-place mat-vec: ST[source leg-step][variable]; rows padded by two doubles so that the
    // 16-byte writes of blocks with consecutive e1 (same e0) land in different LDS banks (row stride 240 words = 16 mod 32
    // would put them on two bank groups only)
    alignas(16) double ST[STR][NMAX + 2];
    alignas(16) double piv[2][NMAX];
    double pd[2];  // the pivot of the published row (its own slot in the row carries d - 1, see the sweeps)
    double u[NMAX], d[NMAX], r[NMAX], col[NMAX];
    double redv[NW], redw[NW];
    double gamma;
    int redi[NW];
    int wcount[NW];
    Rec rec[NW];
    alignas(8) signed char act[MMAX];
    alignas(8) unsigned char slot[MMAX];
    unsigned char flpc[MMAX];  // bit 0: row has already been switched to its other bound once by the block start; bits 1-7 (continuation and
                               // safe variants): how often the single-row iteration has ADDED the row (anti-cycling, HMPC_READD_LIMIT)
    typedef typename std::conditional<(MMAX > 256), unsigned short, unsigned char>::type row_t;
    row_t Wrow[NMAX];  // working-set slot -> constraint row
    double Ep[EP_LDS];  // E = (N_W M N_W')^-1, packed lower triangle: E(i,j), i>=j, at i(i+1)/2 + j (EGLOBAL: in args.e_scratch)
  };
  union {
    Asm a;
    Sol s;
  } u;
};

// Layout of one hand-over slot (KernelArgs::spill): what a fast variant whose working set is full leaves for the continuation
// variant.  Every member of Smem::Sol before Ep has the same offset in all variants of one shape (NMAX, HMAX, NT, NC, BPT) -- QCAP
// only sizes Ep, the last member -- so the bytes go from the fast variant's LDS into the continuation variant's as they are:
//   [0, A_BYTES)            Sol bytes [0, offsetof(ST)): x, x_u, z, w
//   [A_BYTES, E_OFF)        Sol bytes [offsetof(piv), offsetof(Ep)): pivot rows (unused), u, d, r, col, reductions, act, slot, flpc, Wrow
//   [E_OFF, ...)            the packed Schur inverse, q (q + 1) / 2 doubles
//   [stride - M_BYTES, stride)   the 6 x 6 register blocks of M = H^-1, entry (ii, jj) of thread t at (ii GS + jj) NT + t (coalesced)
// q and the iteration count travel in the status word.  The host sizes the stride for the source variant's QCAP (Variant::spill_stride).
template <class SM, int NT, int BPT>
struct SpillLayout {
  typedef typename SM::Sol SolT;
  static constexpr size_t A_BYTES = offsetof(SolT, ST);
  static constexpr size_t B_OFF = offsetof(SolT, piv), B_BYTES = offsetof(SolT, Ep) - offsetof(SolT, piv);
  static constexpr size_t E_OFF = A_BYTES + B_BYTES;
  static constexpr size_t M_BYTES = (size_t)BPT * GS * GS * NT * sizeof(double);
  static constexpr size_t stride_for(int rows) { return (E_OFF + (size_t)rows * (rows + 1) / 2 * sizeof(double) + M_BYTES + 255) / 256 * 256; }
  static_assert(A_BYTES % 8 == 0 && B_OFF % 8 == 0 && B_BYTES % 8 == 0, "copied as 8-byte words");
};

// index of H(i,j), i <= j, in the folded upper-triangle staging array: row i (< NMAX/2) and row NMAX-1-i share one
// storage row of NMAX+1 entries
template <int NMAX>
__device__ __forceinline__ int hs_index(int i, int j) {
  const bool first = 2 * i < NMAX;
  return (first ? i : NMAX - 1 - i) * (NMAX + 1) + (first ? j - i : j + 1);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a binary64 value every lane agrees on, moved to scalar registers
__device__ __forceinline__ double uni_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// a decision every lane agrees on (computed from broadcast LDS reads), told to the compiler as a scalar
__device__ __forceinline__ bool ub(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// ---- wave-level primitives on binary64 via DPP (no LDS traffic) ----
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_xor1(double v) { return dpp_move<0xB1>(v); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ double dpp_xor2(double v) { return dpp_move<0x4E>(v); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// minimum over the 64 lanes, returned in every lane
__device__ __forceinline__ double wave_min(double v) {
  v = dmin(v, dpp_move<0xB1>(v));   // lane ^ 1
  v = dmin(v, dpp_move<0x4E>(v));   // lane ^ 2
  v = dmin(v, dpp_move<0x141>(v));  // row_half_mirror: joins the two quads of each 8
  v = dmin(v, dpp_move<0x140>(v));  // row_mirror: joins the two halves of each row of 16
  const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
  return dmin(dmin(r0, r1), dmin(r2, r3));
}

// body rotation from the quaternion (RobotState.cpp:17-30; Eigen toRotationMatrix closed form) and its transpose
__device__ inline void quat_to_R(const float *q, float *R, float *Rt) {
  const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1.0f - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0f - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0f - (txx + tyy);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) Rt[k * 3 + i] = R[i * 3 + k];
}

}  // namespace hmpc
// Waves per SIMD ("execution unit") each variant is compiled for -- the second argument of HIP's __launch_bounds__ --
// which fixes the register budget (512 / waves):
//   fast variants with h <= 10 (256 threads / 120 variables / working set <= 64 rows: 49 KB LDS; 128 threads / 60
//   variables: 26 KB LDS): THREE waves per SIMD = 168 VGPRs = three resp. six resident workgroups per CU.  The kernel is
//   bound by dependent-instruction latency and barrier phases, not by issue: the third wave is worth +30 % (profiles/r02).
//   Everything else (h = 20 scratch, safe variants, 512 threads): two.
// Tuning constants.  (Until round 6 this was a list of 29 `#ifndef HMPC_*` A/B switches; every one whose other side had been
// measured worse -- profiles/r02 ... r05 keep the numbers -- is now simply the code: FLIP4, PIN_SWEEP, CHAIN_BALANCE,
// MFS_PUBLISH_FIRST, S0_ACTIVE_ROWS, REFINE_FROM_X, MFS_DEAL_PAIRED, MFS_RCP_NEWTON = 2, MFS_PST_PAD = 18, SCHUR_MFMA*, BLOCK_FRICTION,
// MFMA_SWEEP*, and the wrong-numbers timing switches MFS_NO_* / MFS_ONLY_WAVE.  What is left as a macro is compiled with a
// non-default value by tests/test_switches_compile.py so that it cannot rot.)
namespace hmpc {
constexpr int WAVES_PER_EU_256 = 3, WAVES_PER_EU_128 = 3;  // fast variants: three waves per SIMD (168 VGPRs) where the LDS allows it
constexpr int BLOCK_ROUNDS_2C = 2;     // block start: rounds at most, 120-variable variants (a third instantiation: 86 spilled registers)
constexpr int BLOCK_ROUNDS_3C = 3;     // ... three-contact variant
constexpr int BLOCK_MIN_NEW_2C = 3;    // a further round needs at least this many newly violated rows (5 until the Schur matrix went to the matrix cores: profiles/r05/block_round_ab.txt)
constexpr int BLOCK_MIN_NEW_3C = 2;    // ... three-contact variant (its single-row iteration is dearer)
constexpr int EPT_3C = 7;              // three-contact variant on 256 threads: packed-triangle entries per thread in the register-resident Schur inversion
constexpr int MFS_GT = 4;              // matrix-core sweeps, 20 tiles per wave: tiles per group of operand reads (two groups in flight)
constexpr int MFS_PST_PAD = 18;        // padding of a pivot-panel row in doubles (see MfsPanel; 16 was the round-4 layout: 4-way bank conflicts on the column publishes)
}  // namespace hmpc
#ifndef HMPC_EARLY_HANDOVER_MARGIN
#define HMPC_EARLY_HANDOVER_MARGIN 8  // fast variants: candidates beyond the block start's capacity that send an instance to the continuation variant at once
#endif
#ifndef HMPC_READD_LIMIT
#define HMPC_READD_LIMIT 6  // continuation / safe variants: single-row additions of ONE row before it is set aside (0: off)
#endif
#ifndef HMPC_CONT_ITER_BUDGET
#define HMPC_CONT_ITER_BUDGET 128  // continuation variant: iterations a resumed solve may add before it is left to the safe pass
#endif
#ifndef HMPC_CONT_ROUNDS
#define HMPC_CONT_ROUNDS 4       // continuation variant: block rounds a resumed solve runs before its single-row iteration (0 .. 4)
#endif
namespace hmpc {

// A per-thread constant derived from threadIdx.x.  LAZY = false: an ordinary register.  LAZY = true (the two-blocks-per-thread
// variants, whose 144 registers of matrix leave no room for long-lived scalars): recomputed at every use from an opaque
// copy of the thread index -- one or two integer instructions instead of a register that lives, and spills, for the whole solve.
template <bool LAZY, class F>
struct LazyInt {
  int v;
  F f;
  __device__ __forceinline__ operator int() const {
    if constexpr (LAZY) {
      int t = (int)threadIdx.x;
      asm volatile("" : "+v"(t));
      return f(t);
    } else {
      return v;
    }
  }
};
template <bool LAZY, class F>
__device__ __forceinline__ LazyInt<LAZY, F> lazy_int(F f) {
  return LazyInt<LAZY, F>{LAZY ? 0 : f((int)threadIdx.x), f};
}

// ================================================================================================================
// Stage S on the binary64 matrix cores (round 4; the fast 120-variable / 256-thread variants and the fast three-contact variant).
//
// M = H^-1 by symmetric Gauss-Jordan sweeps with 4 x 4 BLOCK pivots: for the pivot set K (four consecutive variables),
//     D = A_KK;   A_KK <- -D^-1;   A_Kj <- D^-1 A_Kj;   A_iK <- A_iK D^-1;   A_ij <- A_ij - A_iK D^-1 A_Kj      (after n/4 steps A = -H^-1)
// so that the update of the whole matrix is ONE rank-4 product per step -- v_mfma_f64_16x16x4_f64 on 16 x 16 tiles.  The
// symmetric matrix lives in the accumulator registers as its 36 upper-triangle tiles (128 x 128 with identity padding), nine
// tiles = 72 VGPRs per lane, exactly the footprint of the scalar sweeps' 6 x 6 block.  As in the scalar sweeps the rows and
// columns of K take the same fused update with substituted multipliers: the published pivot panel P = A_K,: carries D - I in
// the columns of K, then  A_Kj - (I - D^-1) A_Kj = D^-1 A_Kj,  A_iK - A_iK D^-1 (D - I) = A_iK D^-1,  and the pivot block comes
// out as 2I - D^-1: its diagonal is patched by -2.
// Per step: the wave that owns the pivot's diagonal tile publishes the panel rows AND the inverse of the 4 x 4 pivot block
// (rows of D^-1 by LDL' in pivot order: backward stable for the positive definite block); after the barrier every lane forms its A operand
// -(D^-1 P)[g][16 I + c] (four multiply-adds per tile row), reads its B operand P[g][16 J + c] and issues nine matrix
// instructions; tiles that hold the next panel publish it.  One barrier per step, 30 steps for 120 variables instead of 120.
// The code is specialised per wave (WV is a template parameter): tile coordinates are compile-time constants, LDS addresses
// immediate offsets, and the pivot's tile row is dispatched by ONE computed jump per step -- conditional tests per tile cost
// more than the arithmetic (scripts/micro/sweep_mfma64.hip has the stand-alone measurements: 78 k cycles per workgroup at three
// per CU against 122 k for the scalar sweeps, once the workgroups are out of lockstep as they are in this kernel).
// In the product kernel (profiles/r04/phase_cycles.txt): stage S -- tile load, sweeps, hand-over -- 108 k against 123 k + 3.5 k of
// block loading, but the workgroups sharing the SIMDs slow down by ~15 % in their latency-bound phases (a binary64 matrix
// instruction holds the double-precision pipe for 64 cycles at a time; wave priorities do not change that): +1.3 % end to
// end on the 2-contact h = 10 workload, +5 % at h = 20.
// Precision: an explicitly inverted 4 x 4 pivot block carries cond(D) into the update, which sequential scalar pivots do not --
// unscaled, forces came out up to 6e-5 from qpOASES at 10x the nominal input ranges (7e-8 with scalar pivots).  H_ii spans
// 2e-4 .. 500, so the matrix is scaled first: H~ = 2^k H 2^k, k_i = -floor(log2 H_ii / 2) -- powers of two, exact in both
// directions -- which brings the 10x case back to 1.5e-7 and leaves nominal inputs where they were (5.7e-8).  The pivot block
// itself is factorised by LDL' in pivot order (closed-form 2 x 2 determinants lost another two digits) from its raw entries
// (not recovered from the panel's D - I).
// Afterwards M is handed to the rest of the kernel in the layout everything downstream is built on -- 6 x 6 leg-step
// blocks, one or two per thread -- through passes over an LDS staging area (rows of M in chunks of 48).
// Three contacts (180 variables, 12 x 12 tiles, 20 per wave = 160 VGPRs next to two blocks = 144 VGPRs per thread): tiles and
// blocks do not fit the register file together, which shapes both hand-overs -- see mfs_park_blocks, mfs_owner, mfs_move_blocks.
// 203 k cycles against 265 k for the scalar sweeps there.
typedef double hmpc_d4 __attribute__((ext_vector_type(4)));

// tile t (block-row-major over I <= J of the NTG x NTG grid of 16 x 16 tiles) -> I, J
constexpr int mfs_tile_i(int t, int ntg) {
  int i = 0, base = 0;
  while (t >= base + (ntg - i)) base += ntg - i, ++i;
  return i;
}
constexpr int mfs_tile_j(int t, int ntg) {
  int i = 0, base = 0;
  while (t >= base + (ntg - i)) base += ntg - i, ++i;
  return i + (t - base);
}
// Which wave holds tile (I, J).  By default the NTG (NTG + 1) / 2 tiles are dealt in contiguous runs of the block-row-major
// order (a wave then needs few distinct A operands: one per tile row it touches).  The 12 x 12 grid on four waves (180
// variables, two leg-step blocks per thread) is dealt by hand instead: there the 160 accumulator registers and the 144
// registers of the two blocks have to pass each other in the register file when M is handed over chunk by chunk
// (mfs_relayout), and what bounds the peak is how many of a wave's tiles are still unstored when its blocks are born -- a
// thread's slot-0 block lies in row chunk 0 (wave 3: 0-1), its slot-1 block in chunk 1 / 1-2 / 2-3 / 3 for waves 0..3.  With
// tiles per row chunk (9, 7, 4, 0), (8, 8, 4, 0), (8, 5, 3, 3), (8, 4, 4, 3) for waves 0..3 no wave holds more than 176 registers
// of matrix at any time (contiguous runs: 232, and the allocator spills); every wave owns three diagonal tiles.
constexpr int mfs_owner(int ntg, int nwv, int I, int J) {
  if (ntg == 12 && nwv == 4) {
    switch (I) {
      case 0: return J <= 8 ? 0 : 1;
      case 1: return J <= 5 ? 1 : 2;
      case 2: return J <= 3 ? 2 : 3;
      case 3: return J <= 9 ? 0 : 1;
      case 4: return J <= 9 ? 1 : 2;
      case 5: return J <= 7 ? 2 : 3;
      case 6: return J <= 9 ? 0 : 1;
      case 7: return J <= 8 ? 1 : 2;
      case 8: return 3;
      case 9: return 2;
      default: return 3;
    }
  }
  // 8 x 8 grid on four waves (120 variables): tile rows dealt in pairs I, 7 - I (8 + 1, 7 + 2, 6 + 3, 5 + 4 tiles): every wave
  // touches exactly two tile rows (two A operands per step instead of up to four) and owns two diagonal tiles
  if (ntg == 8 && nwv == 4) return I < 4 ? I : 7 - I;
  const int ntiles = ntg * (ntg + 1) / 2, base = ntiles / nwv, rem = ntiles % nwv;
  int t = 0;  // index of (I, J) in block-row-major order
  for (int i = 0; i < I; ++i) t += ntg - i;
  t += J - I;
  int w = 0, first = 0;
  while (w < nwv - 1 && t >= first + base + (w < rem ? 1 : 0)) first += base + (w < rem ? 1 : 0), ++w;
  return w;
}
constexpr int mfs_count(int ntg, int nwv, int wv) {
  int c = 0;
  for (int i = 0; i < ntg; ++i)
    for (int j = i; j < ntg; ++j) c += (mfs_owner(ntg, nwv, i, j) == wv) ? 1 : 0;
  return c;
}
template <int NTG, int NWV>
struct MfsGrid {
  static constexpr int NTILES = NTG * (NTG + 1) / 2;
  static constexpr int TPW = (NTILES + NWV - 1) / NWV;  // accumulator tiles per wave (some waves may hold one less)
};
template <int NTG, int NWV>
using MfsAcc = hmpc_d4[MfsGrid<NTG, NWV>::TPW];  // a wave's accumulator tiles
template <int NTG, int NWV, int WV>
struct MfsTiles {  // the wave's tiles, sorted by (I, J)
  static constexpr int TPW = MfsGrid<NTG, NWV>::TPW;
  int cnt;
  int i[TPW], j[TPW];
  constexpr MfsTiles() : cnt(0), i{}, j{} {
    for (int ii = 0; ii < NTG; ++ii)
      for (int jj = ii; jj < NTG; ++jj)
        if (mfs_owner(NTG, NWV, ii, jj) == WV) i[cnt] = ii, j[cnt] = jj, ++cnt;
    for (int t = cnt; t < TPW; ++t) i[t] = i[cnt - 1], j[t] = j[cnt - 1];  // (a slot beyond the wave's count is never used)
  }
};
static_assert(mfs_count(12, 4, 0) == 20 && mfs_count(12, 4, 1) == 20 && mfs_count(12, 4, 2) == 19 && mfs_count(12, 4, 3) == 19, "12 x 12 deal");
static_assert(mfs_count(8, 4, 0) == 9 && mfs_count(8, 4, 3) == 9, "8 x 8 deal");
template <int NTG>
struct MfsPanel {
  // panel row stride in doubles.  Round 4 used 16 NTG + 16 (= 0 mod 32 banks): the four rows of a B-operand read hit different
  // banks, but the column-tile publishes -- four lanes per 16-lane group writing the four panel ROWS at one column -- were 4-way
  // bank conflicts (LDS conflict cycles 11 % -> 22 % of the LDS-active cycles).  MFS_PST_PAD = 18 gives a row
  // stride of 4 mod 32 banks: those writes are conflict-free, a B-operand read costs one extra LDS cycle (two of its 32 lanes
  // meet on a bank).
  static constexpr int PST = 16 * NTG + MFS_PST_PAD;
  double P[2][4][PST];   // pivot panel rows, double buffered; the K columns carry D - I
  double Dinv[2][4][4];  // inverse of the pivot block
  double Draw[4][4];     // the pivot block itself, as it is (recovering D from the panel's D - I would cost the small pivots --
                         // down to 1e-4 -- three digits; the scalar sweeps pass d beside the row for the same reason)
  int bad;               // mfs_steps<.., CHECK = true>: a pivot of some block came out <= MFS_PIVOT_MIN (matrix not positive definite)
};
constexpr double MFS_PIVOT_MIN = 1e-9;  // on the power-of-two-equilibrated matrix (diagonal in [1, 4))
// leg-step block t (block-row-major over e0 <= e1 of the NG x NG grid of 6 x 6 blocks) -> e0
constexpr int mfs_block_row(int t, int ng) {
  int e = 0, base = 0;
  while (e < ng - 1 && t >= base + (ng - e)) base += ng - e, ++e;
  return e;
}

// power-of-two Jacobi scaling: k_i from the exponent of H_ii (the staging holds binary32 values: exponent field bits 23-30).
// hdiag(i): H_ii of sweep-order variable i < n.  (The caller's barrier follows.)
template <int NTG, class HDiag>
__device__ __forceinline__ void mfs_scale_exponents(const int n, HDiag hdiag, signed char *kexp) {
  const int tid = threadIdx.x;
  if (tid < 16 * NTG) {
    int k = 0;
    if (tid < n) {
      const int ex = (int)((__float_as_uint(hdiag(tid)) >> 23) & 255u) - 127;  // floor(log2(H_ii)), H_ii > 0
      k = -(ex >> 1);
    }
    kexp[tid] = (signed char)k;
  }
}

// Tiles from the binary32 staging of H, code specialised per wave (the 120-variable variants: the whole matrix is staged at
// once).  hinfo(i): what the staging needs to know about sweep-order variable i (one LDS read); hval(hinfo(i), hinfo(j)):
// H(i, j) from the staging, symmetric in its arguments.
template <int NTG, int NWV, int WV, class HInfo, class HVal>
__device__ __forceinline__ void mfs_load(MfsAcc<NTG, NWV> &acc, const int n, HInfo hinfo, HVal hval, const signed char *kexp) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int TPW = MfsGrid<NTG, NWV>::TPW;
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
  // Two rounds of LDS reads, each issued back to back: first what the lane's four rows per tile row and its one column per
  // tile ARE (horizon step and component, or the reference-order index), then the entries themselves; the address
  // arithmetic in between is branch-free.
  int cinf[TPW], rinf[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int j = 16 * T.j[t] + c;
    cinf[t] = hinfo(j < n ? j : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (t == 0 || T.i[t] != T.i[t - 1]) {  // compile time
        const int i = 16 * T.i[t] + g + 4 * r;
        rinf[t][r] = hinfo(i < n ? i : 0);
      } else {
        rinf[t][r] = rinf[t - 1][r];
      }
    }
  }
  float hv[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[t][r] = hval(rinf[t][r], cinf[t]);  // (symmetric in its arguments: the lower half of a diagonal tile reads the mirror)
    }
  // scaled while still binary32: two multiplications by powers of two (exact; |H| <= 1e3 and |k| <= 12 keep clear of the
  // binary32 range on both sides), the row factors shared by the tiles of a tile row
  float srow[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int j = 16 * T.j[t] + c;
    const float scol = __uint_as_float((unsigned)(127 + (int)kexp[j]) << 23);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T.i[t] + g + 4 * r;
      if (t == 0 || T.i[t] != T.i[t - 1]) srow[r] = __uint_as_float((unsigned)(127 + (int)kexp[i]) << 23);  // compile time
      acc[t][r] = (i < n && j < n) ? (double)((hv[t][r] * srow[r]) * scol) : ((i == j) ? 1.0 : 0.0);  // identity padding
    }
  }
}

// Tiles for the variants whose staging of H holds only part of the matrix at a time (three contacts: two passes over the
// block-diagonals): the 6 x 6 leg-step register blocks are filled pass by pass exactly as for the scalar sweeps, then -- H is
// binary32 data -- parked as floats in LDS, block t of the block-row-major order at 36 t (mfs_park_blocks: common code, the
// blocks die there and never meet the accumulators in the register file), and read into tiles from there (mfs_load_parked:
// per-wave code).  The scaling exponents come from the diagonal blocks.
template <int NV, int BPT, int NTHR, int NTG>
__device__ __forceinline__ void mfs_park_blocks(float *hb, const int n, signed char *kexp, const bool (&own)[BPT], const int (&e0)[BPT],
                                                const int (&e1)[BPT], const bool (&live)[BPT], const double (&a)[BPT][GS][GS]) {
  const int tid = threadIdx.x;
  if (tid >= n && tid < 16 * NTG) kexp[tid] = 0;  // padding
#pragma unroll
  for (int s = 0; s < BPT; ++s) {
    if (own[s]) {
      float *dst = hb + (tid + NTHR * s) * (GS * GS);
#pragma unroll
      for (int ii = 0; ii < GS; ++ii)
#pragma unroll
        for (int jj = 0; jj < GS; jj += 2) {
          float2 v2;
          v2.x = (float)a[s][ii][jj], v2.y = (float)a[s][ii][jj + 1];  // (exact: the blocks were loaded from binary32 values)
          *reinterpret_cast<float2 *>(dst + ii * GS + jj) = v2;
        }
    }
    if (live[s] && e0[s] == e1[s]) {
      // k_i = -floor(log2(H_ii) / 2) from the exponent field of the diagonal entries (H_ii > 0)
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        const int ex = ((__double2hiint(a[s][k][k]) >> 20) & 2047) - 1023;
        kexp[GS * e0[s] + k] = (signed char)(-(ex >> 1));
      }
    }
  }
}
template <int NTG, int NWV, int WV, int NV>
__device__ __forceinline__ void mfs_load_parked(MfsAcc<NTG, NWV> &acc, const int n, const float *hb, const signed char *kexp) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int CNT = mfs_count(NTG, NWV, WV), NGB = NV / GS;
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
  // entry (i, j), i <= j, lies in block (e0, e1) = (i / 6, j / 6) at 36 (e0 NGB - e0 (e0 - 1) / 2 + e1 - e0) + 6 (i % 6) + j % 6: a row
  // part and a column part, each computed once per tile row / tile column
  auto rowpart = [&](const int i) __attribute__((always_inline)) -> int {
    const int e = (i * 171) >> 10, ii = i - GS * e;  // i / 6 for i < 256
    return (e * NGB - ((e * (e - 1)) >> 1) - e) * (GS * GS) + ii * GS;
  };
  auto colpart = [&](const int j) __attribute__((always_inline)) -> int {
    const int e = (j * 171) >> 10, jj = j - GS * e;
    return e * (GS * GS) + jj;
  };
  int rp[4] = {0, 0, 0, 0}, rcp[4] = {0, 0, 0, 0};
  float srow[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const int j = 16 * T.j[t] + c;
    const bool jv = j < n;
    const int cp = colpart(jv ? j : 0);
    const float scol = __uint_as_float((unsigned)(127 + (int)kexp[j]) << 23);
    if (t == 0 || T.i[t] != T.i[t - 1]) {  // compile time
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * T.i[t] + g + 4 * r;
        rp[r] = rowpart(i < n ? i : 0), rcp[r] = colpart(i < n ? i : 0);
        srow[r] = __uint_as_float((unsigned)(127 + (int)kexp[i]) << 23);
      }
    }
    const int crp = (T.i[t] == T.j[t]) ? rowpart(jv ? j : 0) : 0;  // diagonal tiles: the lower half reads the mirror
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T.i[t] + g + 4 * r;
      const bool valid = i < n && jv;
      const int off = (T.i[t] == T.j[t] && i > j) ? crp + rcp[r] : rp[r] + cp;
      const float hv = hb[valid ? off : 0];
      // scaled while still binary32: two multiplications by powers of two (exact)
      acc[t][r] = valid ? (double)((hv * srow[r]) * scol) : ((i == j) ? 1.0 : 0.0);  // identity padding
    }
  }
}

// The NTG * 4 block-pivot steps on the tiles in acc (code specialised per wave).  Callers: a barrier between the last read
// of whatever PN aliases and this call.
template <int NTG, int NWV, int WV, bool CHECK = false>
__device__ __forceinline__ void mfs_steps(MfsPanel<NTG> &PN, MfsAcc<NTG, NWV> &acc, const int n) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int TPW = MfsGrid<NTG, NWV>::TPW, PST = MfsPanel<NTG>::PST, CNT = mfs_count(NTG, NWV, WV);
  static_assert(NTG <= 16, "publish(): one case per tile row");
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
  auto pick = [&](const hmpc_d4 &v, int rr) __attribute__((always_inline)) -> double {  // rr uniform
    const double lo = (rr & 1) ? v[1] : v[0], hi = (rr & 1) ? v[3] : v[2];
    return (rr & 2) ? hi : lo;
  };
  auto rcp1 = [](double d) __attribute__((always_inline)) -> double {  // v_rcp_f64 (2^-24) + Newton steps (two: last bit; one: 2e-15)
    double r = __builtin_amdgcn_rcp(d);
    r = dfma(dfma(-d, r, 1.0), r, r);
    r = dfma(dfma(-d, r, 1.0), r, r);  // (one step measured no faster: profiles/r05/rcp_newton_ab.txt)
    return r;
  };
  // row g of D^-1 for the pivot block just published
  auto publish_dinv = [&](const int s) __attribute__((always_inline)) {
    // x = D^-1 e_g by LDL' with the four pivots taken in order -- for a positive definite block that is backward stable, where
    // closed-form 2 x 2 determinants lose cond(D) eps to cancellation (seen at 10x the nominal input ranges: forces 1e-4 off)
    const double(*D)[4] = PN.Draw;
    const double d00 = D[0][0], d10 = D[1][0], d20 = D[2][0], d30 = D[3][0];
    const double d11 = D[1][1], d21 = D[2][1], d31 = D[3][1], d22 = D[2][2], d32 = D[3][2], d33 = D[3][3];
    const double i0 = rcp1(d00);
    const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
    const double e1 = dfma(-l10, d10, d11), i1 = rcp1(e1);
    const double m21 = dfma(-l20, d10, d21), m31 = dfma(-l30, d10, d31);
    const double l21 = m21 * i1, l31 = m31 * i1;
    const double e2 = dfma(-l21, m21, dfma(-l20, d20, d22)), i2 = rcp1(e2);
    const double m32 = dfma(-l31, m21, dfma(-l30, d20, d32));
    const double l32 = m32 * i2;
    const double e3 = dfma(-l32, m32, dfma(-l31, m31, dfma(-l30, d30, d33))), i3 = rcp1(e3);
    if constexpr (CHECK) {  // (callers whose matrix may be rank deficient: the inherited working set of the block start)
      if (!(d00 > MFS_PIVOT_MIN && e1 > MFS_PIVOT_MIN && e2 > MFS_PIVOT_MIN && e3 > MFS_PIVOT_MIN)) PN.bad = 1;
    }
    double y0 = (g == 0) ? 1.0 : 0.0, y1 = (g == 1) ? 1.0 : 0.0, y2 = (g == 2) ? 1.0 : 0.0, y3 = (g == 3) ? 1.0 : 0.0;
    y1 = dfma(-l10, y0, y1);
    y2 = dfma(-l21, y1, dfma(-l20, y0, y2));
    y3 = dfma(-l32, y2, dfma(-l31, y1, dfma(-l30, y0, y3)));
    const double x3 = y3 * i3;
    const double x2 = dfma(-l32, x3, y2 * i2);
    const double x1 = dfma(-l31, x3, dfma(-l21, x2, y1 * i1));
    const double x0 = dfma(-l30, x3, dfma(-l20, x2, dfma(-l10, x1, y0 * i0)));
    if (c == 0) {
      double *dst = PN.Dinv[s & 1][g];
      dst[0] = x0, dst[1] = x1, dst[2] = x2, dst[3] = x3;
    }
  };
  // panel of step s from the accumulators (rr = s % 4):
  //   row tiles (Ik, J):      lane (g, c) holds A[16 Ik + 4 rr + g][16 J + c] in register rr
  //   column tiles (I < Ik):  lanes with c in [4 rr, 4 rr + 4) hold A[16 I + g + 4 r][16 Ik + c], r = 0..3
  auto publish_ik = [&](auto ikc, const int s, const int rr) __attribute__((always_inline)) {
    constexpr int IK = decltype(ikc)::value;
    if constexpr (IK < NTG) {
      const int c0 = 4 * rr;
      double(*P)[PST] = PN.P[s & 1];
#pragma unroll
      for (int t = 0; t < CNT; ++t) {
        if (T.i[t] == IK) {  // compile time
          double v = pick(acc[t], rr);
          if (T.j[t] == IK) {
            if (c >= c0 && c < c0 + 4) PN.Draw[g][c - c0] = v;
            v -= (c == c0 + g) ? 1.0 : 0.0;
          }
          P[g][16 * T.j[t] + c] = v;
        } else if (T.j[t] == IK) {
          if (c >= c0 && c < c0 + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) P[c - c0][16 * T.i[t] + g + 4 * r] = acc[t][r];
          }
        }
      }
      // the wave that owns the pivot block inverts it AFTER all of its panel stores are issued: their issue overlaps the LDS
      // round trip of the raw block (with the paired deal the owner publishes its whole tile row)
      {
        bool owner = false;  // compile time
#pragma unroll
        for (int t = 0; t < CNT; ++t) owner = owner || (T.i[t] == IK && T.j[t] == IK);
        if (owner) {
          __builtin_amdgcn_s_waitcnt(0xc07f);  // this wave's own LDS writes are visible to it after a wait
          publish_dinv(s);
        }
      }
    }
  };
  auto publish = [&](const int s) __attribute__((always_inline)) {
    const int rr = s & 3;
    switch (s >> 2) {  // uniform: one computed jump
#define HMPC_MFS_CASE(K) case K: publish_ik(std::integral_constant<int, K>(), s, rr); break;
      HMPC_MFS_CASE(0) HMPC_MFS_CASE(1) HMPC_MFS_CASE(2) HMPC_MFS_CASE(3) HMPC_MFS_CASE(4) HMPC_MFS_CASE(5) HMPC_MFS_CASE(6) HMPC_MFS_CASE(7)
      HMPC_MFS_CASE(8) HMPC_MFS_CASE(9) HMPC_MFS_CASE(10) HMPC_MFS_CASE(11) HMPC_MFS_CASE(12) HMPC_MFS_CASE(13) HMPC_MFS_CASE(14)
#undef HMPC_MFS_CASE
      default: publish_ik(std::integral_constant<int, 15>(), s, rr); break;
    }
  };
  publish(0);
  __syncthreads();
  const int nsteps = (n + 3) >> 2;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    const int rr = s & 3, Ik = s >> 2;
    const double(*P)[PST] = PN.P[s & 1];
    const double x0 = PN.Dinv[s & 1][g][0], x1 = PN.Dinv[s & 1][g][1], x2 = PN.Dinv[s & 1][g][2], x3 = PN.Dinv[s & 1][g][3];
    // tile(I,J) -= Q_I' P_J, Q = D^-1 P: A operand -Q[g][16 I + c], B operand P[g][16 J + c].
    // groups of GT tiles, the operands of group k+1 read while the matrix instructions of group k run
    constexpr int GT = (TPW % 3 == 0) ? 3 : (TPW < MFS_GT ? TPW : MFS_GT), NGRP = (CNT + GT - 1) / GT;
    double aop[2][GT], bop[2][GT];
    double alast = 0.0;
    auto fetch = [&](const int grp, double (&ao)[GT], double (&bo)[GT]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int t = grp * GT + u;
        if (t < CNT) {  // compile time
          bo[u] = P[g][16 * T.j[t] + c];
          if (t == 0 || T.i[t] != T.i[t - 1]) {  // compile time; the wave's tiles are sorted by I
            const int m = 16 * T.i[t] + c;
            alast = -dfma(x3, P[3][m], dfma(x2, P[2][m], dfma(x1, P[1][m], x0 * P[0][m])));
          }
          ao[u] = alast;
        }
      }
    };
    fetch(0, aop[0], bop[0]);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
      if (grp + 1 < NGRP) fetch(grp + 1, aop[(grp + 1) & 1], bop[(grp + 1) & 1]);
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int t = grp * GT + u;
        if (t < CNT) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[grp & 1][u], bop[grp & 1][u], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < CNT; ++t)
      if (T.i[t] == T.j[t] && T.i[t] == Ik) {  // (first test compile time, second uniform): the pivot block's diagonal
        asm volatile("");
        const double two = (c == 4 * rr + g) ? 2.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] -= (r == rr) ? two : 0.0;
      }
    if (s + 1 < nsteps) publish(s + 1);
    __syncthreads();
  }
}

// M = -A (scaling undone) into the 6 x 6 leg-step blocks, BPT per thread: rows of M staged in chunks of 48 -- three tile rows
// = eight leg-steps, so that which tiles store in a pass is known at compile time and no block straddles a chunk -- with an
// odd row stride.  stage: 48 * (NV + 1) doubles of LDS that nothing else uses until the function returns.
// With two blocks per thread the accumulators and the blocks do not fit the register file together (160 + 144): a slot's block
// is only born at the first chunk it can lie in (known per wave at compile time: thread t's slot s holds block t + NTHR s of the
// block-row-major order), by which time the tiles of the earlier chunks are stored and dead.
template <int NTG, int NWV, int WV, int NV, int BPT, int NTHR>
__device__ __forceinline__ void mfs_relayout(double *stage, MfsAcc<NTG, NWV> &acc, const int n, const signed char *kexp,
                                             const int (&e0)[BPT], const int (&e1)[BPT], const bool (&live)[BPT], double (&a)[BPT][GS][GS]) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int CNT = mfs_count(NTG, NWV, WV);
  constexpr int RC = 48, SST = NV + 1, NCH = (NTG + 2) / 3, NGB = NV / GS, NBLK = NGB * (NGB + 1) / 2;
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    // a slot's block is born (zero) at the first chunk it can lie in -- after that chunk's tiles are stored and dead
    auto birth = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < BPT; ++s) {
        const int b_lo = NTHR * s + 64 * WV;  // compile time: first block a thread of this wave can hold in slot s
        const int pf = (b_lo < NBLK) ? mfs_block_row(b_lo, NGB) / 8 : 0;
        if (p == pf) {
#pragma unroll
          for (int ii = 0; ii < GS; ++ii)
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) a[s][ii][jj] = 0.0;
        }
      }
    };
    if (RC * p < n) {  // uniform
#pragma unroll
      for (int t = 0; t < CNT; ++t)
        if (T.i[t] / 3 == p) {  // compile time
          const int j = 16 * T.j[t] + c;
          if (T.j[t] < NTG - 1 || c < NV - 16 * (NTG - 1)) {  // (first test compile time; columns >= NV are padding and are not staged)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(16 * (T.i[t] - 3 * p) + g + 4 * r) * SST + j] = -acc[t][r];
          }
        }
      __syncthreads();
      birth();
#pragma unroll
      for (int s = 0; s < BPT; ++s) {
        const int b_lo = NTHR * s + 64 * WV, b_hi = (b_lo + 63 < NBLK - 1) ? b_lo + 63 : NBLK - 1;
        const bool can = b_lo < NBLK && p >= mfs_block_row(b_lo < NBLK ? b_lo : 0, NGB) / 8 && p <= mfs_block_row(b_hi, NGB) / 8;  // compile time
        if (can && live[s] && (e0[s] >> 3) == p) {  // this slot's block lies in the chunk: rows 6 e0 .., columns 6 e1 ..
          const double *rowp = stage + (GS * e0[s] - RC * p) * SST + GS * e1[s];
          const bool dg = (e0[s] == e1[s]);
          int kr[GS], kc[GS];
#pragma unroll
          for (int k = 0; k < GS; ++k) kr[k] = (int)kexp[GS * e0[s] + k], kc[k] = (int)kexp[GS * e1[s] + k];
#pragma unroll
          for (int ii = 0; ii < GS; ++ii) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) {
              // off-diagonal blocks lie above the diagonal of M: staged as they are; a diagonal block takes its lower
              // triangle from the mirror of the upper one (bit-identical halves, as the scalar path leaves them)
              const int off = (ii > jj) ? (dg ? jj * SST + ii : ii * SST + jj) : ii * SST + jj;
              a[s][ii][jj] = rowp[off] * __hiloint2double((1023 + kr[ii] + kc[jj]) << 20, 0);  // M = 2^k M~ 2^k (exact)
            }
            // (two blocks per thread: keep the scheduler from forming all 36 scale factors ahead of the reads -- 72 registers
            // that the dying tiles and the blocks do not leave)
            if constexpr (BPT > 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    } else {
      birth();
    }
  }
}

// The blocks leave a wave's arm of the switch through real register moves: without them the allocator ties the blocks of the
// four arms (and of the code behind the switch) to the same registers, and the arms -- whose tiles die in different orders --
// cannot all be coloured around that choice (it spilled a tile inside the step loop of one arm).
template <int BPT>
__device__ __forceinline__ void mfs_move_blocks(double (&dst)[BPT][GS][GS], const double (&src)[BPT][GS][GS]) {
#pragma unroll
  for (int s = 0; s < BPT; ++s)
#pragma unroll
    for (int ii = 0; ii < GS; ++ii)
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) asm volatile("v_mov_b64 %0, %1" : "=v"(dst[s][ii][jj]) : "v"(src[s][ii][jj]));
}

// Stage S in one call for the variants whose staging holds the whole of H at once and whose threads hold one block each (120
// variables on 256 threads: 8 x 8 tiles on four waves): scaling, tiles, steps, hand-over.  (The same code for 60 variables on
// 128 threads -- 4 x 4 tiles on two waves -- was built and measured in round 4: one workgroup's stage 20 % shorter, the kernel
// 1.4 % SLOWER with six workgroups per CU behind each other's 64-cycle matrix instructions; removed in round 5,
// profiles/r04/mfma_sweep_experiments.txt keeps the numbers.)
template <int NTG, int NWV, int WV, int NV, int NT, class HInfo, class HVal>
__device__ __forceinline__ void mfma_sweeps(MfsPanel<NTG> &PN, double *stage, const int n, HInfo hinfo, HVal hval, signed char *kexp, const int e0,
                                            const int e1, const bool live, double (&a)[1][GS][GS]) {
  mfs_scale_exponents<NTG>(n, [&](const int i) __attribute__((always_inline)) { const int inf = hinfo(i); return hval(inf, inf); }, kexp);
  __syncthreads();
  MfsAcc<NTG, NWV> acc;
  mfs_load<NTG, NWV, WV>(acc, n, hinfo, hval, kexp);
  __syncthreads();  // every tile is loaded before the panel (which aliases the staging of H) is written
  mfs_steps<NTG, NWV, WV>(PN, acc, n);
  const int e0a[1] = {e0}, e1a[1] = {e1};
  const bool la[1] = {live};
  mfs_relayout<NTG, NWV, WV, NV, 1, NT>(stage, acc, n, kexp, e0a, e1a, la, a);
}


// ---- The Schur matrix of the block start on the same machinery (round 5) -----------------------------------------------------
// S0 = N_W M N_W' (k0 <= 16 NTG rows, packed lower triangle in LDS) is inverted by the 4 x 4 block-pivot steps above instead of
// two scalar pivots per barrier on a packed triangle spread over the threads' registers: (k0 + 3) / 4 barrier-separated steps
// instead of k0 / 2, and the update is one matrix instruction per 16 x 16 tile instead of ~12 binary64 instructions per entry
// and thread.  Same power-of-two equilibration as stage S (the diagonal of S0: ~50 for a moment / line-contact row, ~1e4 for a friction row).
template <int NTG>
struct SchurPanel {
  MfsPanel<NTG> pn;
  signed char kexp[16 * NTG];
};
// (the caller's barrier follows)
template <int NTG>
__device__ __forceinline__ void schur_scale_exponents(const int k0, const double *Ep, signed char *kexp) {
  const int tid = threadIdx.x;
  if (tid < 16 * NTG) {
    int k = 0;
    if (tid < k0) {
      const int ex = ((__double2hiint(Ep[(unsigned)(tid * (tid + 1) / 2 + tid)]) >> 20) & 2047) - 1023;  // floor(log2 S0_ii); S0_ii > 0
      k = -(ex >> 1);
    }
    kexp[tid] = (signed char)k;
  }
}
template <int NTG, int NWV, int WV>
__device__ __forceinline__ void schur_load(MfsAcc<NTG, NWV> &acc, const int k0, const double *Ep, const signed char *kexp) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int CNT = mfs_count(NTG, NWV, WV);
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
  // the scaling exponents straight from the diagonal of S0 (what schur_scale_exponents leaves in kexp for the store at the end):
  // no barrier between that pass and this one
  auto kof = [&](const int i) __attribute__((always_inline)) -> int {
    if (i >= k0) return 0;
    const int ex = ((__double2hiint(Ep[(unsigned)(i * (i + 1) / 2 + i)]) >> 20) & 2047) - 1023;
    return -(ex >> 1);
  };
  (void)kexp;
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const int j = 16 * T.j[t] + c;
    const int kj = kof(j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T.i[t] + g + 4 * r;
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const bool valid = hi < k0;
      const double v = Ep[(unsigned)(valid ? hi * (hi + 1) / 2 + lo : 0)];
      acc[t][r] = valid ? v * __hiloint2double((1023 + kof(i) + kj) << 20, 0) : ((i == j) ? 1.0 : 0.0);  // identity padding
    }
  }
}
// E = S0^-1 = -A (scaling undone) back into the packed triangle (callers: a barrier before E is read)
template <int NTG, int NWV, int WV>
__device__ __forceinline__ void schur_store(const MfsAcc<NTG, NWV> &acc, const int k0, double *Ep, const signed char *kexp) {
  constexpr MfsTiles<NTG, NWV, WV> T;
  constexpr int CNT = mfs_count(NTG, NWV, WV);
  const int ln = threadIdx.x & 63, g = ln >> 4, c = ln & 15;
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const int j = 16 * T.j[t] + c;
    const int kj = (int)kexp[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * T.i[t] + g + 4 * r;
      if (i <= j && j < k0)  // (upper triangle of the tile grid: i <= j always holds off the diagonal tiles)
        Ep[(unsigned)(j * (j + 1) / 2 + i)] = -acc[t][r] * __hiloint2double((1023 + (int)kexp[i] + kj) << 20, 0);
    }
  }
}
// the whole inversion, code specialised per wave; returns through SP.pn.bad whether a pivot was not positive
template <int NTG, int NWV, int WV>
__device__ __forceinline__ void schur_invert(SchurPanel<NTG> &SP, const int k0, double *Ep) {
  MfsAcc<NTG, NWV> acc;
  schur_load<NTG, NWV, WV>(acc, k0, Ep, SP.kexp);
  mfs_steps<NTG, NWV, WV, true>(SP.pn, acc, k0);
  schur_store<NTG, NWV, WV>(acc, k0, Ep, SP.kexp);
}

// three waves per SIMD = 3 (256 threads) or 6 (128 threads) workgroups per CU: their LDS must fit the CU's 160 KB
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT>
constexpr bool fits_three_waves() {
  return BPT == 1 && sizeof(Smem<NMAX, HMAX, NT, QCAP, NC, BPT>) * (size_t)(3 * 256 / NT) <= (size_t)160 * 1024;
}

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// Which 6 x 6 leg-step blocks of the symmetric matrix a thread holds in registers: thread t owns blocks number t, t + NT, ... (< NG (NG + 1) / 2)
// of the sweep order, block-row-major: (e0, e1), e0 <= e1.  BPT == 1: plain registers.  BPT == 2 (256 VGPRs, 144 of them the two
// blocks): the coordinates of a slot travel PACKED in one register (bits 0-5 e0, 6-11 e1, 12 owner) and are unpacked where they are
// used, behind an opaque copy (fence) so that the compiler cannot hoist the unpacked values back into registers that live -- and
// spill -- for the rest of the kernel: fence() at the head of a phase makes what the phase unpacks live for that phase only.
template <int NG, int NT, int BPT>
struct BlockOwner {
  static constexpr int NTILE = NG * (NG + 1) / 2;
  int e0_r[BPT], e1_r[BPT], i0_r[BPT], j0_r[BPT];
  bool owner_r[BPT], diag_r[BPT];
  unsigned pk[BPT];
  __device__ __forceinline__ void fence() {
    if constexpr (BPT == 2) {
#pragma unroll
      for (int s = 0; s < BPT; ++s) asm volatile("" : "+v"(pk[s]));
    }
  }
  __device__ __forceinline__ int E0(const int s) const { if constexpr (BPT == 1) return e0_r[s]; else return (int)(pk[s] & 63u); }
  __device__ __forceinline__ int E1(const int s) const { if constexpr (BPT == 1) return e1_r[s]; else return (int)((pk[s] >> 6) & 63u); }
  __device__ __forceinline__ int I0(const int s) const { if constexpr (BPT == 1) return i0_r[s]; else return GS * (int)(pk[s] & 63u); }
  __device__ __forceinline__ int J0(const int s) const { if constexpr (BPT == 1) return j0_r[s]; else return GS * (int)((pk[s] >> 6) & 63u); }
  __device__ __forceinline__ bool OWN(const int s) const { if constexpr (BPT == 1) return owner_r[s]; else return ((pk[s] >> 12) & 1u) != 0; }
  __device__ __forceinline__ bool DIAG(const int s) const {
    if constexpr (BPT == 1) return diag_r[s];
    else { const unsigned v = pk[s]; return (v & 63u) == ((v >> 6) & 63u); }
  }
  // (called right before the first load of the blocks: nothing of it is live during the chains of H)
  __device__ __forceinline__ void assign() {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < BPT; ++s) {
      const int t = tid + s * NT;
      int ea = 0;
      while (ea < NG - 1 && (ea + 1) * NG - (ea + 1) * ea / 2 <= t) ++ea;
      const bool ow = t < NTILE;
      const int eb = ow ? ea + (t - (ea * NG - ea * (ea - 1) / 2)) : 0;
      ea = ow ? ea : 0;
      if constexpr (BPT == 1) {
        owner_r[s] = ow, e1_r[s] = eb, e0_r[s] = ea;
        diag_r[s] = (ea == eb);
        i0_r[s] = GS * ea, j0_r[s] = GS * eb;
      } else {
        pk[s] = (unsigned)ea | ((unsigned)eb << 6) | (ow ? 4096u : 0u);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Stage A1 + A2 of the kernel as a function.  In: the instance's packed record in S.u.a.rec (stage A0), the robot constants in
// args.  Out, in LDS: the trigonometric tables, x0, Acd, Bcd, the weights W and the per-step constraint block Fc (Smem::Asm);
// the per-contact constraint normals Cn, the stance prefix counts and every index table of the swing elimination, ub7 / sc7, n, m,
// nls (Smem's persistent part); the identity power Apow[0].  Ends with a barrier.
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT>
__device__ __forceinline__ void stage_a_scalars(Smem<NMAX, HMAX, NT, QCAP, NC, BPT> &S, const KernelArgs &args, const int inst, const int h, Prof &prof) {
  using SM = Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  using RL = RecLayout<NC>;
  constexpr int NG = SM::NG, U = SM::U, PS = SM::PS, C8 = 8 * NC;
  auto &A = S.u.a;
  const int tid = threadIdx.x;
  (void)prof;
  const float *rf = reinterpret_cast<const float *>(A.rec);
  const unsigned char *gait = reinterpret_cast<const unsigned char *>(A.rec + RL::NF + 12 * h);
  const float *in_p = rf + RL::P, *in_v = rf + RL::V, *in_q = rf + RL::Q, *in_w = rf + RL::W, *in_r = rf + RL::R,
              *in_ja = rf + RL::JA, *in_wt = rf + RL::WT;
  // Fz cap of a contact: f_max for the feet; the hand's own cap travels in the extension record
  auto fz_cap = [&](int c) __attribute__((always_inline)) -> float { return (NC == 3 && c == 2) ? rf[RL::FMH] : args.f_max; };
  // ---------------- A1: trigonometry, one lane per angle (SolverMPC.cpp:374-393, 333-342, 74-85); a lane of another
  // wave builds the swing-leg elimination tables meanwhile (SolverMPC.cpp:589-637)
  {
    const double PI = 3.14159265359, PI2 = 2 * PI;
    constexpr int L_ROLL = (NT >= 256) ? 65 : 65, L_PITCH = (NT >= 256) ? 128 : 66, L_YAW = (NT >= 256) ? 192 : 12;
    auto joint = [&](int i) __attribute__((always_inline)) -> float {
      float a = in_ja[i];
      const int k = i % 5;
      if (k == 2 || k == 4) a = (float)((double)a + 0.3 * PI);
      if (k == 3) a = (float)((double)a - 0.6 * PI);
      const double ad = (double)a;
      return (float)((__builtin_fabs(ad) < PI2) ? ad : fmod(ad, PI2));  // fmod(x,y) == x exactly when |x| < y
    };
    if (tid < 12) {
      // lanes 0..9: the joint angles; lanes 10, 11: q2+q3+q4 of each leg -- one sincos evaluation for all twelve
      const int b = 5 * (tid - 10);
      const float ang = (tid < 10) ? joint(tid) : (joint(b + 2) + joint(b + 3)) + joint(b + 4);
      double s, c;
      det_sincos((double)ang, s, c);
      float *dst = (tid < 10) ? A.sc[tid] : A.sc234[tid - 10];
      dst[0] = (float)s;
      dst[1] = (float)c;
    } else if (tid == L_ROLL || tid == L_PITCH || tid == L_YAW) {
      // the three Euler angles are the longest scalar chains of the stage (two binary64 divisions, a square root and a
      // sincos each): one lane each in different waves, so that they run side by side instead of one after the other
      const float qw = in_q[0], qx = in_q[1], qy = in_q[2], qz = in_q[3];
      if (tid == L_ROLL) {
        float n0 = 2.0f * (qw * qx + qy * qz);
        double d0 = 1.0 - (double)(2.0f * (qx * qx + qy * qy));
        A.rpy[0] = (float)det_atan2((double)n0, d0);
      } else if (tid == L_PITCH) {
        float t = qw * qy - qx * qz;
        double asd = 2.0 * (double)t;
        if (!(asd < 0.99999)) asd = 0.99999;
        float as = (float)asd;
        float pitch = (float)det_asin((double)as);
        A.rpy[1] = pitch;
        double s, c;
        det_sincos((double)pitch, s, c);
        A.ypsc[2] = (float)c;
        A.ypsc[3] = (float)s;
      } else {
        float n2 = 2.0f * (qw * qz + qx * qy);
        double d2 = 1.0 - (double)(2.0f * (qy * qy + qz * qz));
        float yaw = (float)det_atan2((double)n2, d2);
        A.rpy[2] = yaw;
        double s, c;
        det_sincos((double)yaw, s, c);
        A.ypsc[0] = (float)c;
        A.ypsc[1] = (float)s;
      }
    } else if (tid == 64) {
      // a leg-step survives iff its Fz bound f_max*gait is not ~0 (SolverMPC.cpp:589-637): per-step prefix counts here,
      // the index tables are filled in parallel in the next stage
      int nv = 0, nl = 0;
      for (int i = 0; i < h; ++i) {
        int bits = 0, nst = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float ubc = fz_cap(c) * (float)gait[NC * i + c];
          const bool st = !(ubc < 0.0001 && ubc > -.0001);
          bits |= st ? (1 << c) : 0;
          nst += (int)st;
        }
        A.pre_nl[i] = (unsigned char)(nl > 255 ? 255 : nl);
        A.pre_nv[i] = (unsigned char)(nv > 255 ? 255 : nv);
        A.pre_st[i] = (unsigned char)bits;
        nv += 6 * nst;
        nl += nst;
      }
      S.n = nv;
      S.nls = nl;
      S.m = 8 * nl;
    }
    // constant structure of Acd (identity), Bcd (zeros) and the constraint block (zeros), filled by everyone
    for (int t = tid; t < 169; t += NT) A.Acd[t] = (t % 14 == 0) ? 1.0f : 0.0f;  // fl(delta + dt*0) = delta
    for (int t = tid; t < PS; t += NT) A.Bcd[t] = 0.0f;                           // fl(dt*0) = 0
    for (int t = tid; t < C8 * U; t += NT) A.Fc[t] = 0.0f;
    for (int t = tid; t < U * h; t += NT) S.rmap[t] = 255;
    if (tid == 0) S.relax = args.relax, S.pad0 = 0;  // (pad0: "a row keeps being re-added", the anti-cycling variants' flag)
  }
  __syncthreads();
  PROF_MARK(P_A1);

  // ---------------- A2: scalar algebra (RobotState.cpp:17-47, SolverMPC.cpp:65-89,302-331,420-433,488-548): body on a
  // lane of wave 0, one foot per lane of two other waves (different waves, so the three run concurrently)
  if (tid == 0) {
    float R[9], Rt[9];
    quat_to_R(in_q, R, Rt);
    float Rbi[9];
    {
      const float cy = A.ypsc[0], sy = A.ypsc[1], cp = A.ypsc[2], sp = A.ypsc[3];
      float Rb[9] = {cy * cp, -sy, 0.0f, sy * cp, cy, 0.0f, -sp, 0.0f, 1.0f};
      inverse3(Rb, Rbi);
    }
    for (int i = 0; i < 3; ++i) {
      A.x0[i] = A.rpy[i];
      A.x0[3 + i] = in_p[i];
      A.x0[6 + i] = in_w[i];
      A.x0[9 + i] = in_v[i];
    }
    A.x0[12] = args.gravity;                                     // 9.81f in the reference (SolverMPC.cpp:420)
    const float Ib[3] = {args.Ib[0], args.Ib[1], args.Ib[2]};  // 0.5413, 0.5200, 0.0691 (RobotState.cpp:45)
    float RI[9], Iw[9], Iinv[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) RI[i * 3 + k] = R[i * 3 + k] * Ib[k];
    chain_mm<3, 3, 3>(RI, Rt, Iw);
    inverse3(Iw, Iinv);

    // continuous model -> forward Euler (SolverMPC.cpp:312-331, 145-146); mass: hmpc_params (9.0 at :423)
    const float dt = args.dt;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A.Acd[i * 13 + 6 + j] = 0.0f + dt * Rbi[i * 3 + j];
    for (int i = 0; i < 3; ++i) A.Acd[(3 + i) * 13 + 9 + i] = 0.0f + dt * 1.0f;
    A.Acd[11 * 13 + 12] = 0.0f + dt * -1.0f;
    const float inv_m = args.inv_mass;  // fl(1.0f / mass), mass = 9.0 in the reference (SolverMPC.cpp:423)
    for (int leg = 0; leg < NC; ++leg) {
      const float r0 = in_r[0 * NC + leg], r1 = in_r[1 * NC + leg], r2 = in_r[2 * NC + leg];
      float cm[9] = {0.0f, -r2, r1, r2, 0.0f, -r0, -r1, r0, 0.0f};
      float blk[9];
      chain_mm<3, 3, 3>(Iinv, cm, blk);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A.Bcd[(6 + i) * U + 3 * leg + j] = dt * blk[i * 3 + j];
          A.Bcd[(6 + i) * U + 3 * NC + 3 * leg + j] = dt * Iinv[i * 3 + j];
        }
      for (int i = 0; i < 3; ++i) A.Bcd[(9 + i) * U + 3 * leg + i] = dt * inv_m;
    }
    for (int s = 0; s < 12; ++s) A.W[s] = in_wt[s];
    A.W[12] = 0.0f;
  }
  {
    // foot rotation Rz(q0)Rx(q1)Ry(q2)Ry(q3)Ry(q4) and this leg's 8 rows of the 16x12 constraint block
    // (SolverMPC.cpp:426-433, 488-548)
    const int leg_lane0 = (NT >= 256) ? 128 : 64, leg_lane1 = (NT >= 256) ? 192 : 65, leg_lane2 = (NC == 3) ? (NT >= 512 ? 256 : 96) : -1;
    if (tid == leg_lane0 || tid == leg_lane1 || tid == leg_lane2) {
      const int leg = (tid == leg_lane0) ? 0 : (tid == leg_lane1 ? 1 : 2);
      float R[9], Rt[9];
      quat_to_R(in_q, R, Rt);
      const float mu = args.mu_inst ? args.mu_inst[inst] : args.mu;  // 2.0 in the reference (SolverMPC.cpp:488); per instance for terrain sweeps
      const float lt = args.lt, lh = args.lh;                         // 0.09, 0.06 (SolverMPC.cpp:489-490)
      const int b = (leg < 2) ? 5 * leg : 0;
      const float s0 = A.sc[b][0], c0 = A.sc[b][1], s1 = A.sc[b + 1][0], c1 = A.sc[b + 1][1];
      const float s2 = A.sc[b + 2][0], c2 = A.sc[b + 2][1], s3 = A.sc[b + 3][0], c3 = A.sc[b + 3][1];
      const float s4 = A.sc[b + 4][0], c4 = A.sc[b + 4][1];
      const float s234 = A.sc234[leg & 1][0], c234 = A.sc234[leg & 1][1];
      // every operation in the type C++ gives it at SolverMPC.cpp:428-433: the "1.0" / "-1.0" literals promote the
      // product they start, and whatever that product is combined with, to binary64; one narrowing per entry
      const float a = c0 * s2 + (c2 * s0) * s1;
      const double bb = (double)(c0 * c2) - (((double)s0 * (double)s1) * (double)s2);
      const float d = c2 * s0 + (c0 * s1) * s2;
      const double e = (double)(s0 * s2) - (((double)c0 * (double)c2) * (double)s1);
      const double X = (double)(c3 * a) + (double)s3 * bb;
      const double Y = ((double)s3 * (double)a) - (double)c3 * bb;
      const double P = (double)(c3 * d) - ((double)s3) * e;
      const double Q = (double)(s3 * d) + (double)c3 * e;
      float Rf[9];
      Rf[0] = (float)((-(double)s4) * X - (double)c4 * Y);
      Rf[1] = -(c1 * s0);
      Rf[2] = (float)((double)c4 * X - (double)s4 * Y);
      Rf[3] = (float)((double)c4 * P - (double)s4 * Q);
      Rf[4] = c0 * c1;
      Rf[5] = (float)((double)c4 * Q + (double)s4 * P);
      Rf[6] = -(s234 * c1);
      Rf[7] = s1;
      Rf[8] = c234 * c1;
      if (NC == 3 && leg == 2) {  // the hand's contact frame is an input of the extension record
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = rf[RL::RH + k];
      }
      float col0[3], col1[3], vlt[3], vlh[3], t0[3], t1[3], flt[3], flh[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        col0[k] = Rf[k * 3 + 0];
        col1[k] = Rf[k * 3 + 1];
        vlt[k] = -lt * Rf[k * 3 + 2];
        vlh[k] = -lh * Rf[k * 3 + 2];
      }
      chain_mm<1, 3, 3>(col0, Rt, t0);
      chain_mm<1, 3, 3>(col1, Rt, t1);
      chain_mm<1, 3, 3>(vlt, Rt, flt);
      chain_mm<1, 3, 3>(vlh, Rt, flh);
      float *row = A.Fc + (8 * leg) * U;
      const int cf = 3 * leg, cmo = 3 * NC + 3 * leg;
      row[0 * U + cf + 0] = -mu, row[0 * U + cf + 2] = 1.0f;
      row[1 * U + cf + 0] = mu, row[1 * U + cf + 2] = 1.0f;
      row[2 * U + cf + 1] = -mu, row[2 * U + cf + 2] = 1.0f;
      row[3 * U + cf + 1] = mu, row[3 * U + cf + 2] = 1.0f;
      for (int j = 0; j < 3; ++j) {
        row[4 * U + cmo + j] = t0[j];
        row[5 * U + cf + j] = flt[j];
        row[5 * U + cmo + j] = t1[j];
        row[6 * U + cf + j] = flh[j];
        row[6 * U + cmo + j] = (leg != 1) ? -t1[j] : t1[j];
      }
      row[7 * U + cf + 2] = 2.0f;
      // per-leg 8x6 constraint normals in binary64 (columns: F then M of this leg)
      for (int rr = 0; rr < 8; ++rr)
        for (int k = 0; k < 3; ++k) {
          S.Cn[leg][rr][k] = (double)row[rr * U + cf + k];
          S.Cn[leg][rr][3 + k] = (double)row[rr * U + cmo + k];
        }
    }
  }
  if (args.ext_Fc) {  // uniform.  Parity hook: the constraint block comes from outside (after the lanes above have written theirs)
    __syncthreads();
    const float *xf = args.ext_Fc + (size_t)inst * C8 * U;
    for (int t = tid; t < C8 * U; t += NT) A.Fc[t] = xf[t];
    for (int t = tid; t < NC * 8 * 6; t += NT) {
      const int leg = t / 48, rr = (t / 6) % 8, k = t % 6;
      S.Cn[leg][rr][k] = (double)xf[(8 * leg + rr) * U + ((k < 3) ? 3 * leg + k : 3 * NC + 3 * leg + k - 3)];
    }
  }
  {
    // index tables, one work item per (step, leg, k): two orders of the surviving variables --
    //  reference order (ascending original index, SolverMPC.cpp:644-658): used to build H, g bit-identically;
    //  sweep order (leg-step major, [F(3), M(3)] each): used by the solver, 6x6 blocks = leg-step pairs.
    const int nl_tot = S.nls;
    if (tid >= 64 && tid < 128 && nl_tot <= NG) {
      for (int it = tid - 64; it < 3 * NC * h; it += 64) {
        const int i = it / (3 * NC), leg = (it / 3) % NC, k = it % 3;
        const int st = A.pre_st[i];
        if (st & (1 << leg)) {
          const int nst = __popc(st), rank = __popc(st & ((1 << leg) - 1));
          const int e = A.pre_nl[i] + rank, nv = A.pre_nv[i];
          const int oF = nv + 3 * rank + k, oM = nv + 3 * nst + 3 * rank + k;
          S.vstep[oF] = (unsigned char)i, S.vcomp[oF] = (unsigned char)(3 * leg + k);
          S.vstep[oM] = (unsigned char)i, S.vcomp[oM] = (unsigned char)(3 * NC + 3 * leg + k);
          S.o2s[oF] = (unsigned char)(GS * e + k), S.s2o[GS * e + k] = (unsigned char)oF;
          S.o2s[oM] = (unsigned char)(GS * e + 3 + k), S.s2o[GS * e + 3 + k] = (unsigned char)oM;
          S.sinfo[GS * e + k] = (unsigned short)(i | ((3 * leg + k) << 8));
          S.sinfo[GS * e + 3 + k] = (unsigned short)(i | ((3 * NC + 3 * leg + k) << 8));
          S.rmap[U * i + 3 * leg + k] = (unsigned char)(GS * e + k);
          S.rmap[U * i + 3 * NC + 3 * leg + k] = (unsigned char)(GS * e + 3 + k);
          if (k == 0) {
            S.ls_leg[e] = (unsigned char)leg;
            S.ls_step[e] = (unsigned char)i;
            const double u7 = (double)(fz_cap(leg) * (float)gait[NC * i + leg]);
            S.ub7[e] = u7;
            S.sc7[e] = (u7 > 1.0) ? 1.0 / u7 : 1.0;  // the Fz cap is O(f_max): compare it on a unit scale
          }
        }
      }
    }
  }
  // identity power
  for (int t = tid; t < 169; t += NT) A.Apow[t] = (t % 14 == 0) ? 1.0f : 0.0f;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Stages A3 / A4 and the gradient as a function.  In (LDS, Smem::Asm): Acd, Bcd, x0, W, the identity in Apow[0], the record's
// trajectory; the reference-order index tables vstep / vcomp / o2s.  Out: Phi_k = Acd^k Bcd for k < h, the tracking error e
// (Smem::Asm) and g in sweep order (Smem::g, binary64 copy of the binary32 value).  n = reduced variables of the instance.
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT>
__device__ __forceinline__ void stage_a_chains(Smem<NMAX, HMAX, NT, QCAP, NC, BPT> &S, const KernelArgs &args, const int inst, const int h, const int n, Prof &prof) {
  using SM = Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  using RL = RecLayout<NC>;
  constexpr int U = SM::U, PS = SM::PS;
  auto &A = S.u.a;
  const int tid = threadIdx.x;
  (void)prof;
  const float *in_traj = reinterpret_cast<const float *>(A.rec) + RL::NF;
  // ---------------- A3/A4: Acd^k by repeated right-multiplication from the identity (SolverMPC.cpp:148-158), and from each
  // power as it appears: Phi_k = Acd^k Bcd (:161-178), SPhi = fl(w_s Phi) (B'S first, as B'*S*B evaluates left to right),
  // tracking error e_i = Acd^(i+1) x0 - X_d (:457-461, :570).  Only two powers are kept (ping-pong).
  // The fmaf chains run over the structurally non-zero terms only, in ascending index order: a term whose factor is an
  // exact structural zero (fl(dt*0), the off-diagonal zeros of the identity, the zero pattern of Acd^k) adds +-0 to the
  // running sum, which leaves it bit for bit unchanged (HMPC-A1), so the result equals the dense 13-term chain.
  //   Acd  = I + dt A_ct:  column j has its diagonal 1 and  rows 0..2 (j = 6..8),  row j-6 (j = 9..11),  row 11 (j = 12);
  //   Bcd:                 column of a force component c has rows 6..8 and row 9+c, a moment column rows 6..8;
  //   Acd^k:               row s has its diagonal and  columns 6..8 (s < 3),  s+6 (s = 3, 4),  11 and 12 (s = 5),  12 (s = 11).
  // Every lane runs the same four-term chain; a lane with fewer live terms pads with a term whose factor is a
  // structural zero (row 12 / row 0 of Acd, row 0 of Bcd, column 0 of Acd^k), so there is no divergence.
  // The work items (an entry of the next power, of Phi_k, of e) and their operand addresses do not depend on k: they
  // are set up once, so that a step of the loop is 8 LDS reads, 4 fmaf and the store(s) per item.
  constexpr int NITEM = 169 + PS + 13, NTRIP = (NITEM + NT - 1) / NT;
  int it_p[NTRIP][4];            // offsets of the four Acd^k factors inside the 13x13 power
  const float *it_c[NTRIP][4];   // the four right-hand factors
  int it_kind[NTRIP], it_out[NTRIP];
#pragma unroll
  for (int u = 0; u < NTRIP; ++u) {
    const int t = tid + u * NT;
    int row, cs, m0, m1, m2, m3, kind, out;
    const float *cf;
    if (t < 169) {
      const int i = t / 13, j = t % 13, pad = (j == 12) ? 0 : 12;
      const bool w3 = j >= 6 && j < 9;
      row = i, cf = A.Acd + j, cs = 13, kind = 0, out = t;
      m0 = w3 ? 0 : ((j >= 9 && j < 12) ? j - 6 : (j == 12 ? 11 : pad));
      m1 = w3 ? 1 : pad, m2 = w3 ? 2 : pad, m3 = j;
    } else if (t < 169 + PS) {
      const int rem = t - 169, i = rem / U, j = rem % U;
      row = i, cf = A.Bcd + j, cs = U, kind = 1, out = rem;
      m0 = 6, m1 = 7, m2 = 8, m3 = (j < 3 * NC) ? 9 + j % 3 : 0;
    } else {
      const int s = (t < NITEM) ? t - (169 + PS) : 0;
      row = s, cf = A.x0, cs = 1, kind = (t < NITEM) ? 2 : 3, out = s;
      m0 = s;
      m1 = (s < 3) ? 6 : ((s < 5) ? s + 6 : ((s == 5) ? 11 : ((s == 11) ? 12 : 0)));
      m2 = (s < 3) ? 7 : ((s == 5) ? 12 : 0);
      m3 = (s < 3) ? 8 : 0;
    }
    it_p[u][0] = row * 13 + m0, it_p[u][1] = row * 13 + m1, it_p[u][2] = row * 13 + m2, it_p[u][3] = row * 13 + m3;
    it_c[u][0] = cf + m0 * cs, it_c[u][1] = cf + m1 * cs, it_c[u][2] = cf + m2 * cs, it_c[u][3] = cf + m3 * cs;
    it_kind[u] = kind, it_out[u] = out;
  }
  for (int k = 0; k <= h; ++k) {
    const float *Pk = A.Apow + (k & 1) * 169;
    float *Pn = A.Apow + ((k + 1) & 1) * 169;
#pragma unroll
    for (int u = 0; u < NTRIP; ++u) {
      float acc = ffma(Pk[it_p[u][0]], *it_c[u][0], 0.0f);
      acc = ffma(Pk[it_p[u][1]], *it_c[u][1], acc);
      acc = ffma(Pk[it_p[u][2]], *it_c[u][2], acc);
      acc = ffma(Pk[it_p[u][3]], *it_c[u][3], acc);
      const int kind = it_kind[u], out = it_out[u];
      if (kind == 0) {
        if (k < h) Pn[out] = acc;
      } else if (kind == 1) {
        if (k < h) {
          A.Phi[k * PS + out] = acc;
        }
      } else if (kind == 2 && k >= 1) {
        const float xd = (out < 12) ? in_traj[12 * (k - 1) + out] : 0.0f;
        A.e[13 * (k - 1) + out] = acc - xd;
      }
    }
    __syncthreads();
  }

  PROF_MARK(P_ASM);
  // ---------------- A5: g = 2 (B'S) e and H = 2(B'S B + alpha) (SolverMPC.cpp:569-570), indexed in reference order ------
  if (tid < n) {
    const int a = S.vstep[tid], c = S.vcomp[tid];
    float acc = 0.0f;
    for (int i = a; i < h; ++i) {
      const float *sp = A.Phi + (i - a) * PS + c;
      const float *ep = A.e + 13 * i;
#pragma unroll
      for (int s = 0; s < 13; ++s) acc = ffma(A.W[s] * sp[s * U], ep[s], acc);  // (B'S) first: fl(w_s Phi), then the chain
    }
    S.g[S.o2s[tid]] = args.ext_g ? (double)args.ext_g[(size_t)inst * args.ext_ld + tid] : (double)(2.0f * acc);
  }
  PROF_MARK(P_G);
}

// ---------------------------------------------------------------------------------------------------------------
// The epilogue as a function.  In (LDS): the solution x in sweep order, the working set (act / slot / Wrow, multipliers u), rmap (original
// variable -> sweep index, 255 = eliminated), g, the Fz caps.  Out (HBM): forces in the reference's 12h layout with eliminated variables
// exactly 0 (SolverMPC.cpp:720-732), optionally the binary64 copy, the tick-to-tick working set, the status word (code | iterations << 8 |
// |W| << 20), the flag counter / list of the safe pass, the objective through the KKT identity.
template <int NMAX, int HMAX, int NT, int QCAP, int NC, int BPT>
__device__ __forceinline__ void stage_output(Smem<NMAX, HMAX, NT, QCAP, NC, BPT> &S, const KernelArgs &args, const int inst, const int h, const int n,
                                             const int q, const int iters, const int code, const bool capped, const bool slot_consumed) {
  using SM = Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  constexpr int U = SM::U, C8 = 8 * NC;
  auto &Q = S.u.s;
  const int tid = threadIdx.x;
  int t_out = tid;
  if constexpr (BPT == 2) asm volatile("" : "+v"(t_out));  // (or 8 * tid is formed at the top of the kernel, kept for the whole solve, and spilled)
  for (int t = t_out; t < U * h; t += NT) {
    const int rmp = S.rmap[t];
    const double xv = (rmp == 255) ? 0.0 : Q.x[rmp];
    args.forces[(size_t)inst * U * h + t] = (float)xv;
    if (args.x64) args.x64[(size_t)inst * U * h + t] = xv;
  }
  if (args.wset) {
    // the final working set in original row numbering (rows of eliminated leg-steps and failed solves: 0)
    for (int t = tid; t < C8 * h; t += NT) {
      const int st = t / C8, cc = (t % C8) >> 3, rr = t & 7;
      const int rmp = S.rmap[U * st + 3 * cc];
      signed char av = 0;
      if (rmp != 255 && code == S_OK) av = Q.act[8 * (rmp / GS) + rr];
      args.wset[(size_t)inst * C8 * h + t] = av;
    }
  }
  if (tid == 0) {
    if (slot_consumed) args.spill_slot[inst] = -1;  // the hand-over slot is consumed: a later pass over this instance starts cold
    args.status[inst] = (uint32_t)code | ((uint32_t)(iters & 0xfff) << 8) | ((uint32_t)(q & 0xfff) << 20);
    if ((code == S_WORKSET || code == S_MAXITER || code == S_INFEASIBLE || code == S_KKT) && !capped) {
      if (args.flagged) atomicAdd(args.flagged, 1u);
      if (args.flag_count) {
        const unsigned int k = atomicAdd(args.flag_count, 1u);
        if ((int)k < args.flag_cap) args.flag_list[k] = inst;
      }
    }
    if (args.obj64) {
      // objective through the KKT identity  0.5 x'Hx + g'x = 0.5 g'x + 0.5 u'b_W  (H itself was consumed by the sweeps)
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = dfma(0.5 * S.g[i], Q.x[i], o);
      for (int j = 0; j < q; ++j) {
        const int c = Q.Wrow[j], rr = c & 7;
        const double sj = (double)Q.act[c];
        double bnd = (sj > 0) ? 0.0 : ((rr == 4) ? (double)0.01f : (rr == 7 ? S.ub7[c >> 3] : 0.0));
        const double rl = S.relax;  // != 0 only for an HMPC_S_OK_RELAXED answer: its bounds are the perturbed ones (row_lo / row_ub_calc)
        if (rl != 0.0) {
          const double fr = 0.6180339887498949 * (double)(c + 1);
          const double dl = rl * (1.0 + (fr - __builtin_floor(fr)));
          bnd = (sj > 0) ? -dl : bnd + dl * ((rr == 7 && bnd > 1.0) ? bnd : 1.0);
        }
        o = dfma(0.5 * Q.u[j], sj * bnd, o);
      }
      args.obj64[inst] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage A5 (H) as a function.  In (LDS, Smem::Asm): Phi_k, the weights W, alpha (the record), the reference-order index tables.  Out: H in
// binary32 in the staging area Smem::Asm::Hs -- exact, every block of a block-diagonal a prefix of ONE fmaf chain on
// v_mfma_f32_16x16x4_f32 (FULLBLK variants), 16 x 16 tiles over the reduced variables otherwise -- and, unless stage S reads the staging
// itself (TILES_FROM_STAGING: the matrix-core sweeps of the 120-variable variants), the 6 x 6 register blocks `a` filled from it, pass by
// pass where the staging holds only part of the block-diagonals at a time.  Assembly-only kernels dump H instead.  `bo` is assigned here.
template <int NMAX, int HMAX, int NT, int QCAP, bool ASM_ONLY, int NC, int BPT, bool TILES_FROM_STAGING>
__device__ __forceinline__ void stage_h(Smem<NMAX, HMAX, NT, QCAP, NC, BPT> &S, const KernelArgs &args, const int inst, const int h, const int n, const int ng,
                                        BlockOwner<Smem<NMAX, HMAX, NT, QCAP, NC, BPT>::NG, NT, BPT> &bo, double (&a)[BPT][GS][GS]) {
  using SM = Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  using RL = RecLayout<NC>;
  constexpr int NW = SM::NW, U = SM::U, PS = SM::PS;
  constexpr bool MFMA_SWEEP = TILES_FROM_STAGING;
  auto &A = S.u.a;
  const int tid = threadIdx.x, wv = uni(tid >> 6);
  const auto ln = lazy_int<(BPT == 2)>([](int t) { return t & 63; });
  const float *in_al = reinterpret_cast<const float *>(A.rec) + RL::AL;  // alpha: the diagonal of H
  auto E0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.E0(s); };
  auto E1 = [&](const int s) __attribute__((always_inline)) -> int { return bo.E1(s); };
  auto I0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.I0(s); };
  auto J0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.J0(s); };
  auto OWN = [&](const int s) __attribute__((always_inline)) -> bool { return bo.OWN(s); };
  auto DIAG = [&](const int s) __attribute__((always_inline)) -> bool { return bo.DIAG(s); };
  auto own_blocks = [&]() __attribute__((always_inline)) { bo.assign(); };
  if constexpr (SM::FULLBLK) {
    // H on the matrix cores, through the block-Toeplitz structure of B_qp.  With Phi_k = Acd^k Bcd,
    //     H(a,b) = 2 [ sum_{i >= b} Phi_{i-a}' S Phi_{i-b} + alpha delta_ab ]          (U x U block, steps a <= b)
    // and with j = i - b, d = b - a:  H(a, a+d) = 2 [ T_d(h-1-b) + ... ],  T_d(m) = sum_{j = 0..m} Phi_{j+d}' S Phi_j.
    // So every block of one block-diagonal d is a PREFIX of one and the same chain T_d: h chains of h-d steps (55 block
    // steps at h = 10) instead of one chain per block (220), and the bits are those of the contract's chain for every
    // block -- terms in ascending step order, state rows ascending inside a step, started at +0 (the rows of B_qp above the
    // block diagonal, which the dense chain also visits, are exact zeros and bitwise neutral; HMPC-A1).
    // One chain = one 16x16 accumulator tile per (d, ti, tj) (ti, tj: 16-wide tiles of the U x U block: one for two
    // contacts, four for three); K = 12 live state rows = 3 x k4 per step (the weight of row 12 is 0, SolverMPC.cpp:453);
    // after each step the accumulator is the finished block H(h-1-m-d, h-1-m) and goes to the staging area as
    // 2 (acc + alpha delta): four stores per lane at lane-constant offsets from a per-step base, no index arithmetic.
    // (On the diagonal chain the lower triangle of a block is the unmirrored product; the loader reads the upper one.)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int l15 = ln & 15, kq = ln >> 4;
    const float wq0 = A.W[kq], wq1 = A.W[kq + 4], wq2 = A.W[kq + 8];
    constexpr int TB = (U + 15) / 16, UU = U * U;
    constexpr int HB_SPARE = SM::HS_BLOCKS * UU;  // the 64 spare words behind the blocks the staging area actually holds
    static_assert(HB_SPARE + 64 <= (int)(sizeof(A.Hs) / sizeof(float)), "spare words of the padding lanes lie inside Hs");
    const int nchain = h * TB * TB;
    struct Chain {
      bool live, av, bv;
      int d, len, ra, cb;
      int off[4], vm[4];  // staging offset of the four rows this lane holds inside a block (or its spare word), 1/0
      float al[4];        // alpha on the diagonal entries of the diagonal chain, else 0
    };
    constexpr bool CHAIN_BALANCE = TB == 1 && SM::HSP == 1 && NW == 4 && HMAX <= 10;
    int chain_lim = nchain, chain_dlo = 0;  // the chains of the current staging pass: idx < chain_lim, diagonals from chain_dlo
    auto setup = [&](int idx, Chain &C) __attribute__((always_inline)) {
      int cidx = idx;
      bool alive = idx < chain_lim;
      if constexpr (CHAIN_BALANCE) {
        // two contacts, one staging pass: chain slots -> block-diagonals so that the waves' loads even out.  Slots w and w+4
        // run together on wave w (0|7, 1|6, 2|5, 3|4: 10, 9, 8, 7 steps), slots 10 and 11 (diagonals 8, 9) follow on waves 2
        // and 3: 10 sequential steps on the longest wave instead of 12 (0|4 then 8).  Which wave runs a chain does not
        // touch its arithmetic.
        const int dd = (idx < 4) ? idx : ((idx < 8) ? 11 - idx : ((idx < 10) ? 99 : idx - 2));
        alive = dd < h;
        cidx = dd;
      }
      C.live = alive;
      const int ci = C.live ? cidx : 0;
      C.d = ci / (TB * TB);
      const int ti = (ci / TB) % TB, tj = ci % TB;
      C.len = C.live ? h - C.d : 0;
      C.ra = 16 * ti + l15, C.cb = 16 * tj + l15;
      C.av = C.live && C.ra < U, C.bv = C.live && C.cb < U;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 16 * ti + 4 * kq + rg;
        const bool ok = C.bv && r < U;
        C.vm[rg] = ok ? 1 : 0;
        C.off[rg] = ok ? r * U + C.cb : HB_SPARE + ln;
        C.al[rg] = (ok && C.d == 0 && r == C.cb) ? in_al[r] : 0.0f;
      }
    };
    auto fetch = [&](const Chain &C, int j, float (&o)[6]) __attribute__((always_inline)) {
      const bool la = C.av && j < C.len, lb = C.bv && j < C.len;
      const float *pa = A.Phi + (la ? (j + C.d) * PS + C.ra : 0) + kq * U;
      const float *pb = A.Phi + (lb ? j * PS + C.cb : 0) + kq * U;
      const float a0 = wq0 * pa[0], a1 = wq1 * pa[4 * U], a2 = wq2 * pa[8 * U];  // S Phi = fl(w_s Phi), rows kq, kq+4, kq+8
      const float b0 = pb[0], b1 = pb[4 * U], b2 = pb[8 * U];
      o[0] = la ? a0 : 0.0f, o[1] = la ? a1 : 0.0f, o[2] = la ? a2 : 0.0f;
      o[3] = lb ? b0 : 0.0f, o[4] = lb ? b1 : 0.0f, o[5] = lb ? b2 : 0.0f;
    };
    auto store = [&](const Chain &C, int m, const f4 &acc) __attribute__((always_inline)) {
      if (!(C.live && m < C.len)) return;              // uniform
      const int sb = h - 1 - m, sa = sb - C.d;          // the block this prefix of the chain is
      const int base = (SM::hs_off(C.d, h) - SM::hs_off(chain_dlo, h) + sa) * UU;  // uniform
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) A.Hs[base * C.vm[rg] + C.off[rg]] = 2.0f * (acc[rg] + C.al[rg]);
    };
    // two chains per pass (independent accumulators hide the MFMA dependency latency); the operands of step j+1 are
    // fetched while the matrix instructions of step j run
    auto run_chains = [&](const int idx_lo, const int idx_hi_in) __attribute__((always_inline)) {
    const int idx_hi = CHAIN_BALANCE ? 12 : idx_hi_in;
    for (int idx = idx_lo + wv; idx < idx_hi; idx += 2 * NW) {
      Chain C0, C1;
      setup(idx, C0);
      setup(idx + NW, C1);
      const int steps = C0.len;  // the second chain of the pass lies on a later block-diagonal: it is not longer
      f4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      float c0[6], c1[6], n0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, n1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fetch(C0, 0, c0);
      fetch(C1, 0, c1);
      for (int j = 0; j < steps; ++j) {
        if (j + 1 < steps) {
          fetch(C0, j + 1, n0);
          fetch(C1, j + 1, n1);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[0], c0[3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[0], c1[3], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[1], c0[4], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[1], c1[4], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[2], c0[5], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[2], c1[5], acc1, 0, 0, 0);
        store(C0, j, acc0);
        store(C1, j, acc1);
#pragma unroll
        for (int q2 = 0; q2 < 6; ++q2) c0[q2] = n0[q2], c1[q2] = n1[q2];
      }
    }
    };  // run_chains
    // Block (e0, e1) of the sweep order = leg-steps (sa, la) <= (sb, lb) (step-major); its entry (ii, jj) is the entry
    // (r, c) = (comp(la, ii), comp(lb, jj)) of the staged U x U block (sa, sb), comp(l, k) = 3 l + k for the force and
    // 3 NC + 3 l + k - 3 for the moment components -- read from the upper triangle in the reference order: when both
    // leg-steps lie in the same horizon step, (r, c) with r > c is read at (c, r) (moment-of-la x force-of-lb entries, and
    // the lower triangle of a diagonal block, which therefore comes out exactly symmetric).
    auto load_blocks = [&](const int dlo, const int dhi) __attribute__((always_inline)) {  // blocks whose block-diagonal d = sb - sa lies in [dlo, dhi)
#pragma unroll
      for (int s = 0; s < BPT; ++s) {
        const bool have = OWN(s) && E1(s) < ng;
        const int sa = have ? (int)S.ls_step[E0(s)] : 0, la = have ? (int)S.ls_leg[E0(s)] : 0;
        const int sb = have ? (int)S.ls_step[E1(s)] : 0, lb = have ? (int)S.ls_leg[E1(s)] : 0;
        const int dd = sb - sa;
        // (a thread's slots that hold no live block are zero-filled in the first pass)
        const bool in_pass = (SM::HSP == 1) ? true : (have ? (dd >= dlo && dd < dhi) : (dlo == 0));
        if (!in_pass) continue;
        const int base = (SM::hs_off(dd, h) - SM::hs_off(dlo, h) + sa) * (U * U);
        const float *bN = A.Hs + (have ? base + 3 * la * U + 3 * lb : 0);  // + compR(ii) * U + compC(jj)
        const float *bT = A.Hs + (have ? base + 3 * lb * U + 3 * la : 0);  // transposed position: + compC(jj) * U + compR(ii)
        const bool same = (sa == sb);
#pragma unroll
        for (int ii = 0; ii < GS; ++ii)
#pragma unroll
          for (int jj = 0; jj < GS; ++jj) {
            constexpr int F = 3 * NC - 3;  // comp(l, k) - 3 l - k for a moment component
            const int cr = (ii < 3) ? ii : F + ii, cc = (jj < 3) ? jj : F + jj;
            const bool sw = (ii >= 3 && jj < 3) ? same : ((ii > jj) ? DIAG(s) : false);
            const float v = sw ? bT[cc * U + cr] : bN[cr * U + cc];
            a[s][ii][jj] = have ? (double)v : 0.0;
          }
      }
    };
    // staging passes: the chains of the pass's block-diagonals, then -- behind a barrier -- the register blocks (or the
    // debug dump) that live on them
#pragma unroll
    for (int hp = 0; hp < SM::HSP; ++hp) {
      const int dlo = SM::hs_dlo(hp) < h ? SM::hs_dlo(hp) : h, dhi = SM::hs_dlo(hp + 1) < h ? SM::hs_dlo(hp + 1) : h;
      chain_dlo = dlo, chain_lim = dhi * TB * TB;
      run_chains(dlo * TB * TB, dhi * TB * TB);
      __syncthreads();
      if (args.ext_H) {  // uniform.  Parity hook: the staged entries of the surviving variables are replaced by the caller's
        const float *xh = args.ext_H + (size_t)inst * args.ext_ld * args.ext_ld;
        for (int t = tid; t < n * n; t += NT) {
          const int i = t / n, j = t % n;
          if (i <= j) {
            const int sa = S.vstep[i], sb = S.vstep[j], dd = sb - sa;
            if (dd >= dlo && dd < dhi)
              A.Hs[(SM::hs_off(dd, h) - SM::hs_off(dlo, h) + sa) * U * U + S.vcomp[i] * U + S.vcomp[j]] = xh[i * args.ext_ld + j];
          }
        }
        __syncthreads();
      }
      if constexpr (ASM_ONLY) {
        using DL = DbgLayout<NMAX, NC>;
        for (int t = tid; t < n * n; t += NT) {
          const int i = t / n, j = t % n, lo = i < j ? i : j, hi = i < j ? j : i;  // reference order, upper triangle
          const int sa = S.vstep[lo], sb = S.vstep[hi], dd = sb - sa;
          if (dd >= dlo && dd < dhi)
            args.dbg_f[DL::H + t] = A.Hs[(SM::hs_off(dd, h) - SM::hs_off(dlo, h) + sa) * U * U + S.vcomp[lo] * U + S.vcomp[hi]];
        }
      } else {
        if (hp == 0) own_blocks();
        if constexpr (!MFMA_SWEEP) load_blocks(dlo, dhi);  // (matrix-core sweeps of the 120-variable variants: the staging is read into 16 x 16 tiles in stage S)
      }
      if (hp + 1 < SM::HSP) __syncthreads();  // the next pass overwrites the staging area
    }
  } else {
    // matrix cores: 16x16 output tiles over the reduced variables, K runs over (step i ascending, state row s ascending).
    // Rows of B_qp above the block diagonal are exact zeros, which are bitwise neutral in an fmaf chain started at +0,
    // so operands before a variable's own step are fed as 0 and the chain starts at the tile's first live step.
    // The weight of state row 12 is 0 (SolverMPC.cpp:453) -> that row is neutral too and K = 12 = 3 x k4.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int nti = (n + 15) >> 4;
    const int ntiles = nti * (nti + 1) / 2;
    const int l15 = ln & 15, kq = ln >> 4;
    const float wq0 = A.W[kq], wq1 = A.W[kq + 4], wq2 = A.W[kq + 8];
    // Two tiles per pass (independent accumulators hide the MFMA dependency latency) and the operands of step i+1 are
    // fetched while the matrix instructions of step i run.  A tile's steps before its first live one only add exact zeros.
    struct Tile {
      bool rav, cbv, live;
      int sa, ca, sb, cc, I, J;
    };
    auto setup = [&](int idx, Tile &T) __attribute__((always_inline)) {
      T.live = idx < ntiles;
      int J = 0;
      while ((J + 1) * (J + 2) / 2 <= idx) ++J;
      T.J = J, T.I = idx - J * (J + 1) / 2;
      const int ra = 16 * T.I + l15, cb = 16 * J + l15;
      T.rav = T.live && ra < n, T.cbv = T.live && cb < n;
      T.sa = T.rav ? S.vstep[ra] : 0, T.ca = T.rav ? S.vcomp[ra] : 0;
      T.sb = T.cbv ? S.vstep[cb] : 0, T.cc = T.cbv ? S.vcomp[cb] : 0;
    };
    auto fetch = [&](const Tile &T, int i, float (&o)[6]) __attribute__((always_inline)) {
      const bool la = T.rav && i >= T.sa, lb = T.cbv && i >= T.sb;
      const float *pa = A.Phi + (la ? (i - T.sa) * PS + T.ca : 0) + kq * U;
      const float *pb = A.Phi + (lb ? (i - T.sb) * PS + T.cc : 0) + kq * U;
      const float a0 = wq0 * pa[0], a1 = wq1 * pa[4 * U], a2 = wq2 * pa[8 * U];  // S Phi = fl(w_s Phi), rows kq, kq+4, kq+8
      const float b0 = pb[0], b1 = pb[4 * U], b2 = pb[8 * U];
      o[0] = la ? a0 : 0.0f, o[1] = la ? a1 : 0.0f, o[2] = la ? a2 : 0.0f;
      o[3] = lb ? b0 : 0.0f, o[4] = lb ? b1 : 0.0f, o[5] = lb ? b2 : 0.0f;
    };
    auto store = [&](const Tile &T, const f4 &acc) __attribute__((always_inline)) {
      if (!T.live) return;
      // alpha enters on the diagonal only: looked up (two dependent LDS reads) by the diagonal lanes of diagonal tiles
      float al[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (T.I == T.J) {  // uniform
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int Rr = 16 * T.I + kq * 4 + rg;
          if (kq * 4 + rg == l15 && Rr < n) al[rg] = in_al[S.vcomp[Rr]];
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int Rr = 16 * T.I + kq * 4 + rg, Cc = 16 * T.J + l15;
        if (Rr <= Cc && Cc < n) A.Hs[hs_index<NMAX>(Rr, Cc)] = 2.0f * (acc[rg] + al[rg]);
      }
    };
    for (int idx = wv; idx < ntiles; idx += 2 * NW) {
      Tile T0, T1;
      setup(idx, T0);
      setup(idx + NW, T1);
      const int istart = S.vstep[16 * T0.J];  // the second tile lies further right: its first live step is not earlier
      f4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      float c0[6], c1[6], n0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, n1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fetch(T0, istart, c0);
      fetch(T1, istart, c1);
      for (int i = istart; i < h; ++i) {
        if (i + 1 < h) {
          fetch(T0, i + 1, n0);
          fetch(T1, i + 1, n1);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[0], c0[3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[0], c1[3], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[1], c0[4], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[1], c1[4], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[2], c0[5], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[2], c1[5], acc1, 0, 0, 0);
#pragma unroll
        for (int q2 = 0; q2 < 6; ++q2) c0[q2] = n0[q2], c1[q2] = n1[q2];
      }
      store(T0, acc0);
      store(T1, acc1);
    }
    __syncthreads();
    if (args.ext_H) {  // uniform.  Parity hook, folded layout
      const float *xh = args.ext_H + (size_t)inst * args.ext_ld * args.ext_ld;
      for (int t = tid; t < n * n; t += NT) {
        const int i = t / n, j = t % n;
        if (i <= j) A.Hs[hs_index<NMAX>(i, j)] = xh[i * args.ext_ld + j];
      }
      __syncthreads();
    }
    if constexpr (!ASM_ONLY) {
    own_blocks();
    // register blocks from the folded upper triangle over the reduced variables (reference order)
    if constexpr (!MFMA_SWEEP)
#pragma unroll
    for (int s = 0; s < BPT; ++s)
#pragma unroll
      for (int ii = 0; ii < GS; ++ii)
#pragma unroll
        for (int jj = 0; jj < GS; ++jj) {
          const int i = I0(s) + ii, j = J0(s) + jj;
          double v = 0.0;
          if (i < n && j < n) {
            const int oi = S.s2o[i], oj = S.s2o[j];
            v = (double)A.Hs[hs_index<NMAX>(oi < oj ? oi : oj, oi < oj ? oj : oi)];
          }
          a[s][ii][jj] = v;
        }
    }
  }
}

// MODE 0: the product path, one workgroup = one independent instance.
// MODE 1: COMMAND SWEEPS (hmpc_solve_command_sweep).  The batch is groups of args.sweep_k consecutive records that share
// everything but the reference trajectory -- state, feet, joints, weights, gait table (ConvexMPCLocomotion.cpp:351-406 builds the
// trajectory from the commands; SolverMPC.cpp:398-447, 488-570 builds A_qp, B_qp, H and the constraint block from the state and
// the gait alone) -- so H and its inverse M are a property of the GROUP.  Two launches of this kernel:
//   phase 0 (args.sweep_phase == 0), one workgroup per group: stages A, H, S on the group's first record, M written to the
//     group's slot in HBM (36 x NT doubles, the register blocks' own layout, coalesced), nothing else;
//   phase 1, one workgroup per INSTANCE (the chip stays as full as for independent solves): stage A on the instance's own
//     record (its own g, the same chains), M read from its group's slot instead of stages H and S (55 % of an independent solve),
//     then stages W and Q as they stand.  Same operands, same instructions: forces and status words are bit-identical to MODE 0's.
//   A record that differs from its group's first one anywhere but in the trajectory is not solved (HMPC_S_SWEEP_MISMATCH).
template <int NMAX, int HMAX, int NT, int QCAP, bool ASM_ONLY, int NC = 2, int BPT = 1, int MODE = 0>
__global__ __launch_bounds__(NT, (NT < 512 && fits_three_waves<NMAX, HMAX, NT, QCAP, NC, BPT>()) ? (NT == 128 ? WAVES_PER_EU_128 : WAVES_PER_EU_256) : 2) void hmpc_kernel(KernelArgs args) {
  using SM = Smem<NMAX, HMAX, NT, QCAP, NC, BPT>;
  constexpr bool SWEEP = (MODE == 1);
  static_assert(!SWEEP || (!ASM_ONLY && BPT == 1 && NC == 2 && QCAP != 0), "command sweeps: the fast two-contact variants");
  using RL = RecLayout<NC>;
  constexpr int NG = SM::NG, NW = SM::NW, U = SM::U, PS = SM::PS, C8 = 8 * NC;
  static_assert(NC == 2 || (NC == 3 && NT * BPT >= 512), "contacts: two feet (reference) or two feet + hand (extension)");
  static_assert(BPT == 1 || BPT == 2, "register blocks per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SM &S = *reinterpret_cast<SM *>(smem_raw);
  auto &A = S.u.a;
  auto &Q = S.u.s;
  // the packed Schur inverse: LDS, or this workgroup's slice of the global scratch
  double *const Ep = [&]() __attribute__((always_inline)) -> double * {
    if constexpr (SM::EGLOBAL) return args.e_scratch + (size_t)blockIdx.x * (size_t)(NMAX * (NMAX + 1) / 2);
    else return S.u.s.Ep;
  }();
  auto Eat = [&](int i, int j) __attribute__((always_inline)) -> double & {
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    return Ep[(unsigned)(hi * (hi + 1) / 2 + lo)];  // (unsigned: a scalar base + 32-bit lane offset when E is in global memory)
  };

  const int tid = threadIdx.x, wv = uni(tid >> 6);
  const auto ln = lazy_int<(BPT == 2)>([](int t) { return t & 63; });  // lane (recomputed at every use in the two-block variants)
  if (!ASM_ONLY && args.list_count && blockIdx.x >= *args.list_count) return;  // device-side safe pass: nothing (more) flagged
  // (uniform by construction -- but when it comes from the index list it arrives through a vector load: told to the compiler, so
  //  that the instance's base addresses are scalar arithmetic instead of register pairs that live for the whole kernel)
  const int inst = uni(ASM_ONLY ? args.dbg_index
                                : ((SWEEP && args.sweep_phase == 0) ? (int)blockIdx.x * args.sweep_k  // the group's first record
                                                                    : (args.index_list ? args.index_list[blockIdx.x] : (int)blockIdx.x)));
  const int h = args.horizon;
  if (inst >= args.batch) return;
  // command sweeps: M comes from (phase 1) or goes to (phase 0) this group's slot
  const bool sweep_prepare = SWEEP && args.sweep_phase == 0, sweep_given = SWEEP && args.sweep_phase != 0;
  double *const sweep_m = SWEEP ? args.sweep_m + (size_t)(inst / (SWEEP ? args.sweep_k : 1)) * (size_t)(GS * GS * NT) : nullptr;
  if (!ASM_ONLY && args.cls) {  // uniform: this instance belongs to another variant's launch
    const int c = args.cls[inst];
    if (c < args.cls_lo || c > args.cls_hi) return;
  }
  PROF_DECL;
  // Hand-over of a full working set (KernelArgs::spill, SpillLayout): the fast 120-variable variants SAVE their state, the safe
  // variants of the same shape (working set = variable count, in LDS) RESUME from it -- they assemble the instance again (index
  // tables, constraint normals, g: cheap and bit-identical), then take M, E and the Goldfarb-Idnani state from the slot instead of
  // running stages H, S and the start
  constexpr bool SHAPE_HANDOVER = !ASM_ONLY && NMAX == 120 && NT == 256 && NC == 2 && BPT == 1 && !SM::EGLOBAL;
  constexpr bool SPILLS = SHAPE_HANDOVER && QCAP < HMPC_QCAP_CONT;      // the fast variants (working set of 64 rows, three per CU)
  // the continuation variant (96 rows, two per CU): takes over what the fast variants hand over, with block rounds of its own (up to
  // its 96 rows at once, the Schur matrix as 6 x 6 tiles on the matrix cores), and flags what outgrows it in turn for the safe variant
  constexpr bool RESUMABLE = SHAPE_HANDOVER && QCAP >= HMPC_QCAP_CONT && QCAP < NMAX;
  constexpr bool CONT = RESUMABLE;
  using SPL = SpillLayout<SM, NT, BPT>;
  bool resumed = false;
  if constexpr (RESUMABLE) {
    // (the slot must be this instance's own and its status word must still say "working set full": both are written by the fast
    //  variant in the same solve; anything else -- a stale entry of an earlier batch -- starts cold)
    if (args.resume) resumed = ub(args.spill_slot[inst] == inst && inst < args.spill_cap && (args.status[inst] & 0xffu) == (uint32_t)S_WORKSET);
    if (args.resume == 2 && !resumed) return;  // a continuation-only launch: everything else on the list is the safe variant's
  }
  // the safe-pass variants (working set = variable count, scalar sweeps): they also answer a Hessian that is not positive definite
  // the way the reference's qpOASES run does (KernelArgs::reg_step)
  constexpr bool REGULARISES = !ASM_ONLY && (SM::EGLOBAL || (QCAP >= NMAX && NMAX >= 120));
  if constexpr (REGULARISES) {
    if (args.reg_step) {  // a regularisation step: only the instances the step before it left for this one
      const uint32_t c0 = args.status[inst] & 0xffu;
      if (c0 != (uint32_t)(args.reg_step == 1 ? S_INDEFINITE : S_REG_STEP)) return;
    } else if (args.skip_ok) {  // second pass over a list of flagged instances: what the pass before it solved is left alone
      const uint32_t c0 = args.status[inst] & 0xffu;
      if (c0 == (uint32_t)S_OK || (args.skip_ok == 1 && c0 == (uint32_t)S_OK_RELAXED)) return;  // (2: a relaxed answer gets this pass as well)
    }
  }

  // ---------------- A0: one coalesced burst brings the instance's record into LDS ----------------
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(args.records + (size_t)inst * args.stride);
    const int nwords = args.stride >> 2;
    for (int t = tid; t < nwords; t += NT) A.rec[t] = src[t];
  }
  if constexpr (SWEEP) {
    if (sweep_given) {
      // every word of the record but the trajectory must equal the group's first record's (whose M this solve uses)
      const int first = (inst / args.sweep_k) * args.sweep_k;
      const uint32_t *base = reinterpret_cast<const uint32_t *>(args.records + (size_t)first * args.stride);
      const int nfix = RL::NF, ntraj = 12 * args.horizon, ngw = (NC * args.horizon + 3) >> 2;
      const uint32_t *own = reinterpret_cast<const uint32_t *>(args.records + (size_t)inst * args.stride);
      int bad = 0;
      for (int t = tid; t < nfix + ngw; t += NT) {
        const int w = t < nfix ? t : t + ntraj;
        bad |= (own[w] != base[w]) ? 1 : 0;
      }
      if (__syncthreads_or(bad)) {  // uniform
        for (int t = tid; t < 6 * NC * args.horizon; t += NT) args.forces[(size_t)inst * 6 * NC * args.horizon + t] = 0.0f;
        if (tid == 0) args.status[inst] = S_SWEEP_MISMATCH;
        return;
      }
    }
  }
  __syncthreads();
  PROF_MARK(P_A0);
  const float *rf = reinterpret_cast<const float *>(A.rec);
  const unsigned char *gait = reinterpret_cast<const unsigned char *>(A.rec + RL::NF + 12 * h);
  // Fz cap of a contact: f_max for the feet; the hand's own cap travels in the extension record (the assembly-only dump's bounds)
  auto fz_cap = [&](int c) __attribute__((always_inline)) -> float { return (NC == 3 && c == 2) ? rf[RL::FMH] : args.f_max; };

  // ---------------- A1 + A2: trigonometry, scalar algebra, constraint block, elimination tables (stage_a_scalars above)
  stage_a_scalars<NMAX, HMAX, NT, QCAP, NC, BPT>(S, args, inst, h, prof);
  PROF_MARK(P_A2);

  const int n = uni(S.n), m = uni(S.m), ng = uni(S.nls);
  if (ng > NG) {  // uniform
    if (!ASM_ONLY) {
      for (int t = tid; t < U * h; t += NT) args.forces[(size_t)inst * U * h + t] = 0.0f;
      if (args.wset)  // nothing to carry to the next tick from an instance that was not solved
        for (int t = tid; t < C8 * h; t += NT) args.wset[(size_t)inst * C8 * h + t] = 0;
      if (tid == 0) args.status[inst] = S_TOO_LARGE;
    } else if (tid == 0) {
      args.dbg_i[0] = n;
      args.dbg_i[1] = m;
    }
    return;
  }

  // ---------------- A3/A4 + g: powers of Acd, Phi_k, tracking error, gradient (stage_a_chains above)
  stage_a_chains<NMAX, HMAX, NT, QCAP, NC, BPT>(S, args, inst, h, n, prof);
  // ---- register blocks of the sweeps (stage S): thread t owns the 6x6 blocks number t, t + NT, ... (< NG(NG+1)/2) of the
  // symmetric matrix in sweep order, block-row-major: (e0, e1), e0 <= e1.  Declared here because the blocks are filled
  // straight from the staging area of H, pass by pass where that area holds only part of the block-diagonals at a time.
  // Matrix-core sweeps: the FAST 120-variable variants only.  The safe-pass variants (working set = variable count) keep the
  // scalar sweeps: 4 x 4 block pivots apply an explicitly inverted pivot block, whose forward error carries cond(D) -- at 10x
  // the nominal input ranges that showed as forces up to 9e-5 from qpOASES in the safe pass (7e-8 with scalar pivots), while
  // nominal inputs are unaffected (5.8e-8 either way) and whatever the fast variants get wrong beyond 2e-6 is caught by
  // their KKT check and handed to the safe pass anyway.
  constexpr bool MFMA_SWEEP = !ASM_ONLY && QCAP != 0 && (SM::MFS2 && QCAP < NMAX);
  // (the 60-variable variants are fast-pass only: the safe pass of two-contact batches runs on the 120-variable safe variants)
  // ... and the fast three-contact variant (180 variables, two blocks per thread): 78 tiles, 20 per wave.  Its staging of H holds
  // the block-diagonals in two passes, so the register blocks are filled as for the scalar sweeps and turned into tiles in stage S.
  constexpr bool MFMA_SWEEP3 = SM::MFS3 && !ASM_ONLY && QCAP != 0 && QCAP < NMAX;
  constexpr int MFS3_NTG = (NMAX + 15) / 16;
  constexpr int NTILE = NG * (NG + 1) / 2;
  static_assert(NTILE <= BPT * NT && SM::MMAX <= NT && NMAX <= NT, "threads per block / constraint row / variable");
  // which 6 x 6 register blocks this thread holds (BlockOwner above); thin local names for the accessors
  BlockOwner<NG, NT, BPT> bo;
  auto pk_fence = [&]() __attribute__((always_inline)) { bo.fence(); };
  auto E0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.E0(s); };
  auto E1 = [&](const int s) __attribute__((always_inline)) -> int { return bo.E1(s); };
  auto I0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.I0(s); };
  auto J0 = [&](const int s) __attribute__((always_inline)) -> int { return bo.J0(s); };
  auto OWN = [&](const int s) __attribute__((always_inline)) -> bool { return bo.OWN(s); };
  auto DIAG = [&](const int s) __attribute__((always_inline)) -> bool { return bo.DIAG(s); };
  auto own_blocks = [&]() __attribute__((always_inline)) { bo.assign(); };
  double a[BPT][GS][GS];
  if (!((RESUMABLE && resumed) || sweep_given)) {  // (a resumed solve takes M from its hand-over slot, a command-sweep solve from its group's: no H, no sweeps)
    // ---------------- A5: H = 2(B'SB + alpha) on the matrix cores into the staging area, then the register blocks (stage_h above)
    stage_h<NMAX, HMAX, NT, QCAP, ASM_ONLY, NC, BPT, MFMA_SWEEP>(S, args, inst, h, n, ng, bo, a);
  }  // !resumed

  PROF_MARK(P_HG);
  if (ASM_ONLY) {
    using DL = DbgLayout<NMAX, NC>;
    float *o = args.dbg_f;
    if (tid == 0) {
      args.dbg_i[0] = n;
      args.dbg_i[1] = m;
    }
    for (int t = tid; t < n; t += NT) {
      args.dbg_i[2 + t] = U * S.vstep[t] + S.vcomp[t];
      o[DL::G + t] = (float)S.g[S.o2s[t]];
    }
    for (int t = tid; t < n * n; t += NT) {
      const int i = t / n, j = t % n, lo = i < j ? i : j, hi = i < j ? j : i;  // reference order, upper triangle
      if constexpr (SM::FULLBLK) {
        // (written pass by pass in stage A5)
      } else {
        o[DL::H + t] = A.Hs[hs_index<NMAX>(lo, hi)];
      }
    }
    for (int t = tid; t < C8 * U; t += NT) o[DL::FC + t] = A.Fc[t];
    const float big = 5e10f;
    for (int t = tid; t < C8 * h; t += NT) {
      const int i = t / C8, rr = t % 8, leg = (t % C8) / 8;
      float lbv, ubv;
      if (rr < 4) lbv = 0.0f, ubv = big;
      else if (rr == 4) lbv = 0.0f, ubv = 0.01f;
      else if (rr < 7) lbv = -big, ubv = 0.0f;
      else lbv = 0.0f, ubv = fz_cap(leg) * (float)gait[NC * i + leg];
      o[DL::LB + t] = lbv;
      o[DL::UB + t] = ubv;
    }
    for (int t = tid; t < 13; t += NT) o[DL::X0 + t] = A.x0[t];
    for (int t = tid; t < 169; t += NT) o[DL::ACD + t] = A.Acd[t];
    for (int t = tid; t < PS; t += NT) o[DL::BCD + t] = A.Bcd[t];
    return;
  }

  // =============================== S: M = H^-1 by symmetric sweeps, matrix in registers ===============================
  // Thread t < NG(NG+1)/2 owns the 6x6 block (e, e'), e <= e', of the symmetric matrix in sweep order (block-row-major).
  // Sweep k:  d = a_kk, p = row k;  a_ij -= (p_i/d) p_j  (i,j != k);  a_kj = p_j/d;  a_kk = -1/d.  After n sweeps a = -H^-1.
  // Row k / column k entries take the same fused update with a substituted multiplier (1 - 1/d for row k, d - 1 for
  // column k: p_j - (1-1/d) p_j = p_j/d), so the 6x6 update has no special cases; only a_kk is patched.  The substitution
  // costs nothing: the published pivot row carries d - 1 in the pivot's own slot (d itself travels in Q.pd).
  // The pivot row for sweep k+1 is published to LDS right after sweep k (double buffered) -> one barrier per sweep; the six
  // sweeps of a leg-step are statically unrolled (static register indices).
  const bool is_v = tid < n, is_c = tid < m;
  // (the register blocks were loaded from the staging area of H at the end of stage A5)
  if (sweep_given) {
    if constexpr (SWEEP) {
      // ---- command sweep, phase 1: the group's M as phase 0 left it (the same threads own the same blocks)
      __syncthreads();  // g has been formed from the staging of the assembly, which the solver state aliases
      own_blocks();
#pragma unroll
      for (int ii = 0; ii < GS; ++ii)
#pragma unroll
        for (int jj = 0; jj < GS; ++jj) a[0][ii][jj] = sweep_m[(ii * GS + jj) * NT + tid];
    }
  } else
  if (RESUMABLE && resumed) {
    if constexpr (RESUMABLE) {
      // ---- hand-over: M from the slot's tail (the layout the fast variant's threads left: same block ownership)
      __syncthreads();  // g has been formed from the staging of the assembly, which the solver state aliases
      own_blocks();
      const unsigned char *slotp = args.spill + (size_t)args.spill_slot[inst] * args.spill_stride;
      const double *mb = reinterpret_cast<const double *>(slotp + args.spill_stride - SPL::M_BYTES);
#pragma unroll
      for (int ii = 0; ii < GS; ++ii)
#pragma unroll
        for (int jj = 0; jj < GS; ++jj) a[0][ii][jj] = mb[(ii * GS + jj) * NT + tid];
    }
  } else
  if constexpr (MFMA_SWEEP) {
    // ---- matrix-core sweeps (mfma_sweeps above): tiles from the staging of H, 4 x 4 block pivots, M back in the 6 x 6 blocks
    auto hinfo = [&](const int i) __attribute__((always_inline)) -> int {  // i < n, sweep order
      if constexpr (SM::FULLBLK) return (int)S.sinfo[i];  // horizon step | component << 8
      else return (int)S.s2o[i];                          // reference-order index
    };
    auto hval = [&](const int vi, const int vj) __attribute__((always_inline)) -> float {  // H(i, j) from hinfo(i), hinfo(j)
      if constexpr (SM::FULLBLK) {
        // staged: every U x U block (a <= b) of the unreduced matrix, block (a, a + d) at hs_off(d) + a; a same-step block holds
        // its upper triangle in reference (component) order
        const int si = vi & 255, sj = vj & 255, ri = vi >> 8, rj = vj >> 8;
        const bool sw = (si > sj) || (si == sj && ri > rj);
        const int sa = sw ? sj : si, sb = sw ? si : sj, r = sw ? rj : ri, c = sw ? ri : rj, d = sb - sa;
        return A.Hs[(SM::hs_off(d, h) + sa) * (U * U) + r * U + c];
      } else {
        return A.Hs[hs_index<NMAX>(vi < vj ? vi : vj, vi < vj ? vj : vi)];
      }
    };
    constexpr int NTG1 = (NMAX + 15) / 16;
    // the pivot panels live in LDS that the solver does not use yet: the mat-vec staging
    static_assert(sizeof(MfsPanel<NTG1>) <= sizeof(Q.ST), "room for the pivot panels");
    MfsPanel<NTG1> &PN = *reinterpret_cast<MfsPanel<NTG1> *>(&Q.ST[0][0]);
    double *stage = reinterpret_cast<double *>(&S.u);
    static_assert(sizeof(S.u) / sizeof(double) >= 48 * (NMAX + 1), "re-layout staging of the matrix-core sweeps: 48 rows of M at stride NMAX + 1");
    const bool live0 = bo.owner_r[0] && bo.e1_r[0] < ng;
    static_assert(NW == 4, "per-wave code of the matrix-core sweeps: four waves");
    switch (wv) {  // uniform: per-wave specialised code
      case 0: mfma_sweeps<NTG1, 4, 0, NMAX, NT>(PN, stage, n, hinfo, hval, S.kexp, bo.e0_r[0], bo.e1_r[0], live0, a); break;
      case 1: mfma_sweeps<NTG1, 4, 1, NMAX, NT>(PN, stage, n, hinfo, hval, S.kexp, bo.e0_r[0], bo.e1_r[0], live0, a); break;
      case 2: mfma_sweeps<NTG1, 4, 2, NMAX, NT>(PN, stage, n, hinfo, hval, S.kexp, bo.e0_r[0], bo.e1_r[0], live0, a); break;
      default: mfma_sweeps<NTG1, 4, 3, NMAX, NT>(PN, stage, n, hinfo, hval, S.kexp, bo.e0_r[0], bo.e1_r[0], live0, a); break;
    }
  } else if constexpr (MFMA_SWEEP3) {
    // ---- the same on the tiles filled in stage A5
    MfsPanel<MFS3_NTG> &PN = *reinterpret_cast<MfsPanel<MFS3_NTG> *>(&Q.ST[0][0]);
    static_assert(sizeof(MfsPanel<MFS3_NTG>) <= sizeof(Q.ST), "the pivot panels live in the (not yet used) mat-vec staging");
    double *stage = reinterpret_cast<double *>(&S.u);
    static_assert(sizeof(S.u) / sizeof(double) >= 48 * (NMAX + 1), "re-layout staging of the matrix-core sweeps: 48 rows of M");
    __syncthreads();  // every block is loaded before the staging area of H is written over
    pk_fence();
    int e0a[BPT], e1a[BPT];
    bool lva[BPT], owa[BPT];
#pragma unroll
    for (int s = 0; s < BPT; ++s) e0a[s] = E0(s), e1a[s] = E1(s), owa[s] = OWN(s), lva[s] = OWN(s) && E1(s) < ng;
    float *hb = reinterpret_cast<float *>(&S.u);
    static_assert(sizeof(S.u) >= (size_t)NTILE * GS * GS * sizeof(float), "the parked blocks fit the staging area");
    mfs_park_blocks<NMAX, BPT, NT, MFS3_NTG>(hb, n, S.kexp, owa, e0a, e1a, lva, a);
    __syncthreads();
    switch (wv) {  // uniform: per-wave specialised code
#define HMPC_MFS3_WAVE(W)                                                                  \
  MfsAcc<MFS3_NTG, NW> acc;                                                                \
  mfs_load_parked<MFS3_NTG, NW, W, NMAX>(acc, n, hb, S.kexp);                              \
  __syncthreads(); /* every tile is loaded before the panel (which aliases the parked blocks) is written */ \
  mfs_steps<MFS3_NTG, NW, W>(PN, acc, n);                                                  \
  double aw[BPT][GS][GS];                                                                  \
  mfs_relayout<MFS3_NTG, NW, W, NMAX, BPT, NT>(stage, acc, n, S.kexp, e0a, e1a, lva, aw);  \
  mfs_move_blocks<BPT>(a, aw);
      case 0: { HMPC_MFS3_WAVE(0) } break;
      case 1: { HMPC_MFS3_WAVE(1) } break;
      case 2: { HMPC_MFS3_WAVE(2) } break;
      case 3: { HMPC_MFS3_WAVE(3) } break;
      case 4: { HMPC_MFS3_WAVE((NW > 4 ? 4 : 0)) } break;  // (cases 4-7: the eight-wave wide variant only)
      case 5: { HMPC_MFS3_WAVE((NW > 4 ? 5 : 0)) } break;
      case 6: { HMPC_MFS3_WAVE((NW > 4 ? 6 : 0)) } break;
      default: { HMPC_MFS3_WAVE((NW > 4 ? 7 : 0)) } break;
#undef HMPC_MFS3_WAVE
    }
    pk_fence();
  } else {
  __syncthreads();  // every block is loaded before the solver state (which aliases the staging area) is written
  if constexpr (REGULARISES) {
    if (args.reg_step != 0) {  // uniform
      // H += rho I as qpOASES does after a failed Cholesky factorisation (KernelArgs::reg_step; QProblemB.cpp:1418-1431, 1999-2031):
      // |H|_F over the reduced matrix -- this thread's blocks, an off-diagonal block standing for its mirror image too --,
      // summed in a fixed order (lane partials in LDS, one wave adds them up)
      double ss = 0.0;
#pragma unroll
      for (int s = 0; s < BPT; ++s) {
        double b = 0.0;
#pragma unroll
        for (int ii = 0; ii < GS; ++ii)
#pragma unroll
          for (int jj = 0; jj < GS; ++jj) b = dfma(a[s][ii][jj], a[s][ii][jj], b);
        ss += !OWN(s) ? 0.0 : (DIAG(s) ? b : 2.0 * b);
      }
      double *red = &Q.ST[0][0];
      static_assert(sizeof(Q.ST) >= NT * sizeof(double), "one partial per thread");
      red[tid] = ss;
      __syncthreads();
      if (tid < 64) {
        double t = 0.0;
        for (int k = tid; k < NT; k += 64) t += red[k];
        t += dpp_move<0xB1>(t), t += dpp_move<0x4E>(t), t += dpp_move<0x141>(t), t += dpp_move<0x140>(t);
        t = (readlane_d(t, 0) + readlane_d(t, 16)) + (readlane_d(t, 32) + readlane_d(t, 48));
        if (tid == 0) {
          double rho;
          if (args.reg_step == 1) {
            constexpr double EPS_REG = 1.0e3 * 2.221e-16, SQRT_EPS_REG = 4.7127486671792716e-07;  // Options.cpp:144 (epsRegularisation), Constants.hpp:50
            const double piv = args.reg_rho[inst];  // the pivot that was not positive (reg_step 0 left it here)
            const double er = (piv < 0.0) ? ((-piv + EPS_REG < SQRT_EPS_REG) ? -piv + EPS_REG : SQRT_EPS_REG) : EPS_REG;
            rho = __builtin_sqrt(t) * er;
            args.reg_rho[inst] = rho;
          } else {
            rho = args.reg_rho[inst];
          }
          Q.redv[0] = rho;
        }
      }
      __syncthreads();
      const double rho = uni_d(Q.redv[0]);
#pragma unroll
      for (int s = 0; s < BPT; ++s)
        if (OWN(s) && DIAG(s) && E0(s) < ng) {
#pragma unroll
          for (int ii = 0; ii < GS; ++ii) a[s][ii][ii] += rho;
        }
      if (args.reg_step == 2 && tid < n) {  // the second QP's gradient: g - rho x_1 (QProblem.cpp:1811-1812), x_1 as the force buffer holds it
        const int o = S.s2o[tid];
        S.g[tid] -= rho * (double)args.forces[(size_t)inst * U * h + U * S.vstep[o] + S.vcomp[o]];
      }
    }
  }
  if (tid < NMAX) Q.piv[0][tid] = 0.0, Q.piv[1][tid] = 0.0;
  if constexpr (REGULARISES) {
    if (tid == 0) Q.gamma = 1.0;  // the first sweep pivot that is not positive (free until the solver starts)
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < BPT; ++s)
    if (OWN(s) && E0(s) == 0) {
#pragma unroll
      for (int jj = 0; jj < GS; ++jj)
        if (J0(s) + jj < n) Q.piv[0][J0(s) + jj] = a[s][0][jj];
      if (DIAG(s)) Q.piv[0][0] = a[s][0][0] - 1.0, Q.pd[0] = a[s][0][0];
    }
  __syncthreads();
  pk_fence();
  // waves that hold no register block at all (the second wave of the 128-thread variants: 55 blocks) only keep the barriers
  constexpr int NW_OWN = (NTILE + 63) / 64;
  const bool wave_owns = (NW_OWN >= NW) || (wv < NW_OWN);  // scalar; compile-time true where every wave holds blocks
  for (int kb = 0; kb < ng; ++kb) {
    bool rowb[BPT], colb[BPT], rown[BPT], coln[BPT];
#pragma unroll
    for (int s = 0; s < BPT; ++s) {
      rowb[s] = OWN(s) && (E0(s) == kb);      // my block holds matrix rows 6kb..6kb+5
      colb[s] = OWN(s) && (E1(s) == kb);      // my block holds matrix columns 6kb..6kb+5
      rown[s] = OWN(s) && (E0(s) == kb + 1);  // next leg-step's row / column blocks (publish at the seam)
      coln[s] = OWN(s) && (E1(s) == kb + 1);
    }
#pragma unroll
    for (int kk = 0; kk < GS; ++kk) {
      const int k = kb * GS + kk;
      if (wave_owns) {
      const double *pv = Q.piv[k & 1];
      double *pn = Q.piv[(k + 1) & 1];
      const double d = Q.pd[k & 1];
      if constexpr (REGULARISES) {
        if (!(d > 0.0) && tid == 0 && Q.gamma > 0.0) Q.gamma = d;  // H is not positive definite (rare branch; thread 0 alone reads and writes the slot)
      }
      // v_rcp_f64 is good to 2^-24 (scripts/micro/rcp64_accuracy.hip); one Newton step brings 2e-15, a second one would
      // bring the last bit -- not worth two more fp64 instructions per pivot here: the sweeps' own round-off (cond(H) eps
      // ~ 3e-10) is five orders above it and the substituted multipliers stay consistent with whatever invd is used
      double invd = __builtin_amdgcn_rcp(d);
      invd = dfma(dfma(-d, invd, 1.0), invd, invd);
#pragma unroll
      for (int s = 0; s < BPT; ++s) {
        if (s > 0 && ub(s * NT + wv * 64 >= NTILE)) continue;  // no lane of this wave holds a block in this slot
        double pi[GS], pj[GS];
#pragma unroll
        for (int ii = 0; ii < GS; ii += 2) {
          const double2 t2 = *reinterpret_cast<const double2 *>(pv + I0(s) + ii);
          pi[ii] = t2.x, pi[ii + 1] = t2.y;
        }
#pragma unroll
        for (int jj = 0; jj < GS; jj += 2) {
          const double2 t2 = *reinterpret_cast<const double2 *>(pv + J0(s) + jj);
          pj[jj] = t2.x, pj[jj + 1] = t2.y;
        }
        double qi[GS];
#pragma unroll
        for (int ii = 0; ii < GS; ++ii) qi[ii] = pi[ii] * invd;
        // (the published row carries d - 1 in the pivot's own slot: the threads that hold row k then find q_k = (d-1)/d = 1 - 1/d
        //  and those that hold column k find p_k = d - 1 by themselves -- the substituted multipliers, without any select)
#pragma unroll
        for (int ii = 0; ii < GS; ++ii)
#pragma unroll
          for (int jj = 0; jj < GS; ++jj) a[s][ii][jj] = dfma(-qi[ii], pj[jj], a[s][ii][jj]);
        a[s][kk][kk] = (rowb[s] && colb[s]) ? -invd : a[s][kk][kk];
        // publish row k+1 of the symmetric matrix: (k+1, j >= k+1) from its row blocks, (i < k+1, k+1) from its column blocks
        if (kk + 1 < GS) {
          if (rowb[s]) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj)
              if (!DIAG(s) || jj >= kk + 1) pn[J0(s) + jj] = a[s][(kk + 1) % GS][jj];
            if (DIAG(s)) pn[J0(s) + (kk + 1) % GS] = a[s][(kk + 1) % GS][(kk + 1) % GS] - 1.0, Q.pd[(k + 1) & 1] = a[s][(kk + 1) % GS][(kk + 1) % GS];
          }
          if (colb[s]) {
#pragma unroll
            for (int ii = 0; ii < GS; ++ii)
              if (!DIAG(s) || ii < kk + 1) pn[I0(s) + ii] = a[s][ii][(kk + 1) % GS];
          }
        } else if (kb + 1 < ng) {
          if (rown[s]) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) pn[J0(s) + jj] = a[s][0][jj];
            if (DIAG(s)) pn[J0(s)] = a[s][0][0] - 1.0, Q.pd[(k + 1) & 1] = a[s][0][0];
          }
          if (coln[s] && !DIAG(s)) {
#pragma unroll
            for (int ii = 0; ii < GS; ++ii) pn[I0(s) + ii] = a[s][ii][0];
          }
        }
      }
      // every update of this sweep is complete before the barrier: the pivot-row temporaries die here instead of
      // overlapping the next sweep's reads (the compiler would otherwise sink 30 of the 36 FMAs past the barrier and keep
      // two sets of pivot-row registers alive: +30 VGPRs, the difference between two and three workgroups per CU)
#pragma unroll
      for (int s = 0; s < BPT; ++s)
#pragma unroll
        for (int ii = 0; ii < GS; ++ii)
#pragma unroll
          for (int jj = 0; jj < GS; ++jj) asm volatile("" : "+v"(a[s][ii][jj]));
      }  // wave_owns
      __syncthreads();
    }
  }
  pk_fence();
  if constexpr (REGULARISES) {
    const double negp = uni_d(Q.gamma);  // (the last sweep ended with a barrier)
    if (!(negp > 0.0)) {
      // Not positive definite: nothing a dual active-set method can start from -- x_u = -H^-1 g is not a minimiser; the fast variants
      // (whose matrix-core sweeps do not look at their pivots: a test there costs the headline 0.7-2 %, profiles/r06/NOTES.md) diverge
      // on such an instance and hand it over through their KKT check.  The reference's qpOASES run regularises such a QP
      // (KernelArgs::reg_step has the story); here the instance ends as S_INDEFINITE, its forces zeroed, the offending pivot in
      // reg_rho[inst] for regularisation step 1, and counts as flagged, so that hmpc_download's repair pass picks it up.
      for (int t = tid; t < U * h; t += NT) args.forces[(size_t)inst * U * h + t] = 0.0f;
      if (args.wset)
        for (int t = tid; t < C8 * h; t += NT) args.wset[(size_t)inst * C8 * h + t] = 0;
      if (tid == 0) {
        if (args.reg_rho) args.reg_rho[inst] = negp;
        if (args.reg_count) {  // device-side chain: listed for the regularisation launches that follow
          const unsigned int k = atomicAdd(args.reg_count, 1u);
          if ((int)k < args.reg_cap) args.reg_list[k] = inst;
        }
        args.status[inst] = (uint32_t)S_INDEFINITE;
        if (args.flagged) atomicAdd(args.flagged, 1u);
      }
      return;
    }
  }
  // M = -a.  Diagonal blocks keep the full symmetric 6x6 (their lower triangle is overwritten with the mirror of the upper
  // one, so both halves are bit-identical); off-diagonal blocks hold M(e0,e1) and stand for M(e1,e0) transposed.
#pragma unroll
  for (int s = 0; s < BPT; ++s) {
    const bool own_s = OWN(s), diag_s = DIAG(s);
#pragma unroll
    for (int ii = 0; ii < GS; ++ii)
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) a[s][ii][jj] = own_s ? -a[s][ii][jj] : 0.0;
#pragma unroll
    for (int ii = 1; ii < GS; ++ii)
#pragma unroll
      for (int jj = 0; jj < ii; ++jj) a[s][ii][jj] = diag_s ? a[s][jj][ii] : a[s][ii][jj];
  }
  }  // scalar sweeps
  PROF_MARK(P_SWEEP);
  if constexpr (SWEEP) {
    if (sweep_prepare) {  // uniform.  Phase 0 of a command sweep: this group's M to its slot, nothing else
#pragma unroll
      for (int ii = 0; ii < GS; ++ii)
#pragma unroll
        for (int jj = 0; jj < GS; ++jj) sweep_m[(ii * GS + jj) * NT + tid] = a[0][ii][jj];
      return;
    }
  }

  // ---- products with the register blocks --------------------------------------------------------------------------
  auto blk_rows = [&](const int s, const double (&wj)[GS], double (&ra)[GS]) __attribute__((always_inline)) {  // ra = a[s] * wj   (result on leg-step e0)
#pragma unroll
    for (int ii = 0; ii < GS; ++ii) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int jj = 0; jj < GS; jj += 2) {
        s0 = dfma(a[s][ii][jj], wj[jj], s0);
        s1 = dfma(a[s][ii][jj + 1], wj[jj + 1], s1);
      }
      ra[ii] = s0 + s1;
    }
  };
  auto blk_cols = [&](const int s, const double (&wi)[GS], double (&ca)[GS]) __attribute__((always_inline)) {  // ca = a[s]' * wi  (result on leg-step e1)
#pragma unroll
    for (int jj = 0; jj < GS; ++jj) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int ii = 0; ii < GS; ii += 2) {
        s0 = dfma(a[s][ii][jj], wi[ii], s0);
        s1 = dfma(a[s][ii + 1][jj], wi[ii + 1], s1);
      }
      ca[jj] = s0 + s1;
    }
  };
  // z = M w for a dense w in LDS (entries >= n exactly 0).  Every block writes its row partial to ST[e1][vars of e0]
  // and its mirrored partial to ST[e0][vars of e1]; variable i then sums ST[0..ng-1][i] in index order (deterministic).
  auto rmatvec = [&](const double *w) __attribute__((always_inline)) {
    pk_fence();
    // the partials are staged in STH halves of the source leg-steps (STH = 1: all at once); the running sums keep the
    // same order either way: even sources into s0, odd ones into s1, ascending.  One block at a time, each product formed
    // in the stage that stores it (row product: source e1, column product: source e0) -- nothing is held across a barrier.
    double s0 = 0.0, s1 = 0.0;
    constexpr int HB = NG / 2;
#pragma unroll
    for (int st = 0; st < SM::STH; ++st) {
      const int lo = st * SM::STR;  // sources [lo, lo + STR) in this stage
#pragma unroll
      for (int s = 0; s < BPT; ++s)
        if (OWN(s) && E1(s) < ng) {
          const bool w1 = (SM::STH == 1) || (E1(s) >= lo && E1(s) < lo + SM::STR);
          const bool w0 = (SM::STH == 1) || (E0(s) >= lo && E0(s) < lo + SM::STR);
          // one code path for every block: a diagonal block is stored as the full, exactly symmetric 6x6 and has wi == wj, so
          // its two products are the same bits and its two stores hit the same words with the same values
          if (w1) {
            double wj[GS], ra[GS];
#pragma unroll
            for (int k = 0; k < GS; k += 2) {
              const double2 u2 = *reinterpret_cast<const double2 *>(w + J0(s) + k);
              wj[k] = u2.x, wj[k + 1] = u2.y;
            }
            blk_rows(s, wj, ra);
#pragma unroll
            for (int k = 0; k < GS; k += 2)
              *reinterpret_cast<double2 *>(&Q.ST[E1(s) - lo][I0(s) + k]) = make_double2(ra[k], ra[k + 1]);
          }
          if (w0) {
            double wi[GS], ca[GS];
#pragma unroll
            for (int k = 0; k < GS; k += 2) {
              const double2 t2 = *reinterpret_cast<const double2 *>(w + I0(s) + k);
              wi[k] = t2.x, wi[k + 1] = t2.y;
            }
            blk_cols(s, wi, ca);
#pragma unroll
            for (int k = 0; k < GS; k += 2)
              *reinterpret_cast<double2 *>(&Q.ST[E0(s) - lo][J0(s) + k]) = make_double2(ca[k], ca[k + 1]);
          }
        }
      __syncthreads();
      if (is_v) {
#pragma unroll
        for (int hb = 0; hb < 2 / SM::STH; ++hb) {
          // rows >= ng of ST are never written: they are read all the same (one base address, immediate offsets -- a
          // clamped row index costs one address register per row) and masked out by the selects below
          const int hbase = (SM::STH == 1) ? hb * HB : lo;  // first source of this group of HB rows
          // (BPT == 2: the rows are read in chunks -- 30-40 registers of partials in flight at once, next to two register blocks,
          //  is what the allocator answers with spills; same summation order either way: even sources into s0, odd into s1)
          constexpr int CH = (BPT == 2) ? 6 : HB;
#pragma unroll
          for (int r0 = 0; r0 < HB; r0 += CH) {
            double sv[CH];
#pragma unroll
            for (int r = 0; r < CH; ++r)
              if (r0 + r < HB) sv[r] = Q.ST[((SM::STH == 1) ? hb * HB : 0) + r0 + r][tid];
#pragma unroll
            for (int r = 0; r < CH; ++r)
              if (r0 + r < HB) {
                if (((r0 + r) & 1) == 0) s0 += (hbase + r0 + r < ng) ? sv[r] : 0.0;
                else s1 += (hbase + r0 + r < ng) ? sv[r] : 0.0;
              }
            if constexpr (BPT == 2) asm volatile("" ::: "memory");  // (keeps the chunks' reads from being hoisted together again)
          }
        }
        if (st + 1 == SM::STH) Q.z[tid] = s0 + s1;
      }
      __syncthreads();
    }
  };

  // =============================== solver state ===============================
  // Fixed thread roles: thread c < m = constraint row c (leg-step c>>3, row c&7) with its 6 coefficients and bounds in
  // registers; thread i < n = variable i (leg-step i/6, position i%6).
  const double INF = __builtin_huge_val();
  const double FEAS_TOL = 1e-9;
  constexpr bool LAZY = (BPT == 2);
  // (the fast 256-thread two-contact variants sit exactly on their 168-register budget: their integer roles are recomputed at
  //  every use as well -- one instruction each -- instead of being the allocator's first victims)
  constexpr bool LAZY_IDX = LAZY || (NT == 256 && BPT == 1 && NC == 2 && QCAP != 0 && QCAP < NMAX);
  const auto c_e = lazy_int<LAZY_IDX>([](int t) { return t >> 3; });
  const auto c_rr = lazy_int<LAZY_IDX>([](int t) { return t & 7; });
  // the lower bound of a row is 0 -- except in the last-resort pass (args.relax != 0), where it is recomputed on use
  // rather than kept in a register pair for the whole solve
  // (read through LDS: a value the compiler cannot prove loop-invariant across the barriers, or it hoists the whole
  //  expression back into a register pair that lives -- and spills -- for the rest of the kernel)
  auto row_lo = [&]() __attribute__((always_inline)) -> double {
    const double rl = S.relax;
    if (ub(rl == 0.0)) return 0.0;
    const double fr = 0.6180339887498949 * (double)(tid + 1);
    return -(rl * (1.0 + (fr - __builtin_floor(fr))));
  };
  // upper bound of this thread's row: 0.01 for the moment window, the Fz cap for row 7, else 0 (SolverMPC.cpp:466-482);
  // moved outward in the last-resort pass
  auto row_ub_calc = [&]() __attribute__((always_inline)) -> double {
    const int rr = c_rr;
    double u = (rr == 4) ? (double)0.01f : (rr == 7 ? S.ub7[c_e] : 0.0);
    const double rl = S.relax;
    if (ub(rl != 0.0)) {
      const double fr = 0.6180339887498949 * (double)(tid + 1);
      const double dl = rl * (1.0 + (fr - __builtin_floor(fr)));
      u += dl * ((rr == 7 && u > 1.0) ? u : 1.0);
    }
    return u;
  };
  double c_ub_r = INF;
  bool c_hasl_r = false, c_hasu_r = false;
  const double *c_cn_r = S.Cn[0][0];  // this row's 6 coefficients (LDS; re-read where used: cheaper than 12 live VGPRs)
  if constexpr (!LAZY) {
    if (is_c) {
      const int leg = S.ls_leg[c_e];
      c_cn_r = S.Cn[leg][c_rr];
      c_hasl_r = (c_rr <= 4) || (c_rr == 7);  // every finite lower bound is 0 (SolverMPC.cpp:466-482)
      c_hasu_r = (c_rr >= 4);
      c_ub_r = row_ub_calc();
    }
  }
  // (callers are rows of the QP, tid < m -- except the lane that publishes a selection record when no row is a candidate)
  auto row_cn = [&]() __attribute__((always_inline)) -> const double * {
    if constexpr (!LAZY) return c_cn_r;
    else return (tid < m) ? S.Cn[S.ls_leg[c_e]][c_rr] : S.Cn[0][0];
  };
  auto row_ub = [&]() __attribute__((always_inline)) -> double { if constexpr (!LAZY) return c_ub_r; else return row_ub_calc(); };
  auto row_hasl = [&]() __attribute__((always_inline)) -> bool { if constexpr (!LAZY) return c_hasl_r; else { const int rr = c_rr; return rr <= 4 || rr == 7; } };
  auto row_hasu = [&]() __attribute__((always_inline)) -> bool { if constexpr (!LAZY) return c_hasu_r; else return c_rr >= 4; };
  const auto v_e = lazy_int<LAZY_IDX>([](int t) { return t / GS; });
  const auto v_k = lazy_int<LAZY_IDX>([](int t) { return t % GS; });
  const int v_leg_r = (!LAZY_IDX && is_v) ? S.ls_leg[v_e] : 0;
  auto var_leg = [&]() __attribute__((always_inline)) -> int { if constexpr (!LAZY_IDX) return v_leg_r; else return (tid < n) ? (int)S.ls_leg[v_e] : 0; };
  int q = 0, iters = 0, code = S_OK;
#ifdef HMPC_DEBUG_STATS  // developer build (scripts/dev/cont_probe.py): what happened inside a solve, packed into obj64
  int dbg_bad = 0, dbg_rounds = 0, dbg_norounds_cap = 0, dbg_norounds_few = 0, dbg_dep = 0;
  double dbg_viol = 0.0;  // largest (unit-scaled) violation left when a resumed solve's budget ran out
#define HMPC_DBG(x) x
#else
#define HMPC_DBG(x)
#endif
  if (RESUMABLE && resumed) {
    if constexpr (RESUMABLE) {
      // ---- hand-over: the Goldfarb-Idnani state the fast variant stopped in (x minimises over the working set, u >= 0, E current)
      const uint32_t st0 = args.status[inst];
      q = uni((int)((st0 >> 20) & 0xfffu)), iters = uni((int)((st0 >> 8) & 0xfffu));
      const double *sp = reinterpret_cast<const double *>(args.spill + (size_t)args.spill_slot[inst] * args.spill_stride);
      double *qa = reinterpret_cast<double *>(&Q), *qb = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(&Q) + SPL::B_OFF);
      for (int t = tid; t < (int)(SPL::A_BYTES / 8); t += NT) qa[t] = sp[t];
      for (int t = tid; t < (int)(SPL::B_BYTES / 8); t += NT) qb[t] = sp[SPL::A_BYTES / 8 + t];
      for (int t = tid; t < q * (q + 1) / 2; t += NT) Ep[(unsigned)t] = sp[SPL::E_OFF / 8 + t];
      __syncthreads();
    }
  } else {
  for (int t = tid; t < SM::MMAX; t += NT) {
    Q.act[t] = 0;
    Q.slot[t] = 0;
    Q.flpc[t] = 0;
  }
  if (tid < NMAX) Q.r[tid] = 0.0, Q.u[tid] = 0.0;
  if (tid < NMAX) Q.w[tid] = is_v ? -S.g[tid] : 0.0;  // entries >= n stay exactly 0
  __syncthreads();
  rmatvec(Q.w);  // unconstrained minimiser x_u = -M g
  if (is_v) {
    const double xv = Q.z[tid];
    Q.xu[tid] = xv;
    Q.x[tid] = xv;
  }
  __syncthreads();
  }
  PROF_MARK(P_XU);

  constexpr bool SAFE = SM::EGLOBAL || QCAP >= NMAX;  // the working set cannot overflow
  // ... and of those, the variants hmpc_resolve_failed / the device-side repair launch (always cold): the only ones that ever run
  // for hundreds of iterations (the 60-variable fast variants also have QCAP = NMAX, but are fast variants)
  constexpr bool LONGRUN = SM::EGLOBAL || (QCAP >= NMAX && NMAX >= 120);
  // Anti-cycling (the variants that run the hard instances: continuation and safe).  At a degenerate vertex -- several rows
  // active with zero multipliers -- round-off can make the dual iteration drop and re-add the same rows with zero-length steps
  // until the iteration bound (seen: 600 iterations on the continuation variant, 1 664 on the safe one, each the whole tail of
  // its launch).  In exact arithmetic a row is added at most 3 times per solve in all but the genuinely cycling instances
  // (scripts/dev/emulate_rounds.py, 6x and 10x the input ranges), so a row the single-row iteration has already added
  // HMPC_READD_LIMIT times is set aside like a redundant row: the final KKT check looks at every row again and decides.
  constexpr bool ANTICYCLE = (LONGRUN || CONT) && HMPC_READD_LIMIT > 0;
  const int itmax_v = (SAFE || QCAP >= 140) ? 10 * m + 64 : 4 * m + 16;  // the safe variants may take as long as a cold qpOASES run (nWSR up to ~330 seen)
  int itmax = (args.iter_cap > 0 && args.iter_cap < itmax_v) ? args.iter_cap : itmax_v;
  bool budgeted = false, budget_hit = false;  // (continuation variant only)
  bool perturbed = false;                     // (anti-cycling variants: the bounds were moved outward in this pass)
  const int itmax_full = itmax;
  if constexpr (CONT) {
    // a resumed solve gets a budget of its own: in exact arithmetic the hardest instances need ~50 more changes from the hand-over
    // (scripts/dev/emulate_rounds.py); one that is still going after HMPC_CONT_ITER_BUDGET is cycling at a degenerate vertex -- the
    // safe pass's business (rebuilt E, relaxed bounds), not worth 600 iterations at the tail of this launch.  Such an instance usually
    // SITS at its optimum and trades rows over violations of 1e-9, round-off: when the budget runs out between two iterations the
    // multipliers are refined once and the final KKT check -- every row, relative to the force scale, the standard every solve is
    // held to -- decides between ok and flagged
    if (resumed && iters + HMPC_CONT_ITER_BUDGET < itmax) itmax = iters + HMPC_CONT_ITER_BUDGET, budgeted = true;
  }
  bool c_ignored = false;  // this thread's row was found redundant at a degenerate vertex (violated by round-off only)

  // slack of this thread's constraint row on its tighter side at xv (unit-scaled); side = +1 lower, -1 upper
  auto my_slack = [&](const double *xv, int &side, double &raw) __attribute__((always_inline)) -> double {
    double s0 = 0.0, s1 = 0.0;
    const double *xp = xv + GS * c_e;
    const double *c_cn = row_cn();
#pragma unroll
    for (int k = 0; k < 3; ++k) s0 = dfma(c_cn[k], xp[k], s0);
#pragma unroll
    for (int k = 0; k < 3; ++k) s1 = dfma(c_cn[3 + k], xp[3 + k], s1);
    const double s = s0 + s1;
    const double sl = row_hasl() ? (s - row_lo()) : INF;
    const double su = row_hasu() ? (row_ub() - s) : INF;
    const double ssu = su * ((c_rr == 7) ? S.sc7[c_e] : 1.0);  // (the Fz cap on a unit scale; table built with ub7)
    side = (sl <= ssu) ? 1 : -1;
    raw = (sl <= ssu) ? sl : su;
    return (sl <= ssu) ? sl : ssu;
  };
  // w_i = sum over the active rows of variable i's leg-step of coef(row) * a_row[i]  (+ extra)
  auto gather_w = [&](const double *coefv, double sgn, double extra) __attribute__((always_inline)) {
    if (is_v) {
      const unsigned long long am = *reinterpret_cast<const unsigned long long *>(&Q.act[8 * v_e]);
      const unsigned long long sm = *reinterpret_cast<const unsigned long long *>(&Q.slot[8 * v_e]);
      double acc0 = extra, acc1 = 0.0;
      const int vleg = var_leg();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int ac = (int)(signed char)((am >> (8 * rr)) & 0xff);
        const int sl = (int)((sm >> (8 * rr)) & 0xff);
        const double cf = coefv[sl];
        const double coef = (ac == 0) ? 0.0 : ((ac > 0) ? sgn * cf : -sgn * cf);
        const double cv = S.Cn[vleg][rr][v_k];
        if (rr & 1) acc1 = dfma(coef, cv, acc1);
        else acc0 = dfma(coef, cv, acc0);
      }
      Q.w[tid] = acc0 + acc1;
    }
  };
  // rout[j] (+)= sum_i E(j,i) din[i] for j < q: 4 lanes per row, quad reduction; optional dual ratio test on the result
  auto e_times = [&](const double *din, double *rout, bool accumulate, double &t1c, int &t1j, bool ratio) __attribute__((always_inline)) {
    for (int jb = 0; jb < q; jb += NT / 4) {
      const int j = jb + (tid >> 2), part = tid & 3;
      double acc = 0.0;
      if (j < q) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int i = part; i < q; i += 16) {
          const int i1 = (i + 4 < q) ? i + 4 : i, i2 = (i + 8 < q) ? i + 8 : i, i3 = (i + 12 < q) ? i + 12 : i;
          const double x0 = Eat(j, i), x1 = Eat(j, i1), x2 = Eat(j, i2), x3 = Eat(j, i3);
          const double d0 = din[i], d1 = din[i1], d2 = din[i2], d3 = din[i3];
          a0 = dfma(x0, d0, a0);
          a1 = (i + 4 < q) ? dfma(x1, d1, a1) : a1;
          a2 = (i + 8 < q) ? dfma(x2, d2, a2) : a2;
          a3 = (i + 12 < q) ? dfma(x3, d3, a3) : a3;
        }
        acc = (a0 + a1) + (a2 + a3);
      }
      acc += dpp_xor1(acc);
      acc += dpp_xor2(acc);
      if (j < q && part == 0) {
        if (accumulate) rout[j] += acc;
        else rout[j] = acc;
        if (ratio && acc > 1e-14) {
          const double tj = Q.u[j] / acc;
          if (tj < t1c) t1c = tj, t1j = j;
        }
      }
    }
  };
  // Schur-complement downdate of E when slot l leaves (row/column l are read-only during the pass), then the last slot
  // moves into l.  Two barriers inside.
  // With follow_u the multipliers of the remaining rows follow the removal, u_R <- u_R - E(R,l) u_l / E(l,l)  (= E' d_R).
  auto drop_slot = [&](int l, bool follow_u = false, double ul = 0.0) __attribute__((always_inline)) {
    const double iel = 1.0 / Eat(l, l);
    const int ti = tid >> 4, tj = tid & 15;
    if (follow_u && tid < q && tid != l) Q.u[tid] = dfma(-(Eat(tid, l) * iel), ul, Q.u[tid]);
    for (int ib = 0; ib < q; ib += NT / 16) {
      const int i = ib + ti;
      if (i < q && i != l) {
        const double ci = Eat(i, l) * iel;
        for (int j = tj; j <= i; j += 16) {
          if (j != l) {
            double &ee = Ep[(unsigned)(i * (i + 1) / 2 + j)];
            ee = dfma(-ci, Eat(j, l), ee);
          }
        }
      }
    }
    __syncthreads();
    const int last = q - 1;
    if (l != last) {
      if (tid < last && tid != l) Eat(l, tid) = Eat(last, tid);
      if (tid == l) Eat(l, l) = Eat(last, last);
    }
    if (tid == 0) {
      const int cl = Q.Wrow[l];
      Q.act[cl] = 0;
      if (l != last) {
        const int cm = Q.Wrow[last];
        Q.Wrow[l] = (typename SM::Sol::row_t)cm;
        Q.slot[cm] = (unsigned char)l;
        Q.u[l] = Q.u[last];
      }
    }
    --q;
    __syncthreads();
  };
  // d[slot] = b_j - n_j' x for the active rows (n_j = sign*a_j, b_j = sign*bound)
  auto active_residual = [&](const double *xv) __attribute__((always_inline)) {
    if (is_c) {
      const int ac = Q.act[tid];
      if (ac != 0) {
        double s = 0.0;
        const double *xp = xv + GS * c_e;
        const double *c_cn = row_cn();
#pragma unroll
        for (int k = 0; k < 6; ++k) s = dfma(c_cn[k], xp[k], s);
        const double bnd = (ac > 0) ? row_lo() : row_ub();
        Q.d[Q.slot[tid]] = (double)ac * (bnd - s);
      }
    }
  };

  // =============================== W: block warm start ===============================
  // (the machinery is declared at function scope: the safe variants call a round again from the main loop, as a refresh of E)
    // (a) rows 4-6 (foot-x moment window, toe and heel line contact) violated at x_u take consecutive slots.  The three
    //     rows of a leg-step are linearly independent, rows of different leg-steps touch disjoint variables, so the
    //     Schur matrix of any such set is positive definite.
    // Early hand-over (fast variants that can hand over): a round that finds far more candidate rows than the block start can take
    // (48) -- 8 more than the working set of any nominal instance ever holds -- belongs to an instance that will outgrow the 64-row
    // working set anyway, after dozens of single-row iterations at 11 k cycles each.  It is handed to the continuation variant
    // (96 rows per round, two per CU) right after the rounds instead.  Never at nominal inputs (|W| <= 51 there); at 6x the ranges
    // it takes ~30 iterations off the fast pass for four instances in ten.
    bool early_handover = false;
    double raw = INF;
    int side = 1;
    bool take = false;
    bool tick = false;  // the candidates come from the previous tick's working set (any of the 8 rows of a leg-step)
    int base = 0, k0 = 0, below = 0;
    auto count_candidates = [&]() __attribute__((always_inline)) {
      const unsigned long long bal = __ballot(take);
      below = __popcll(bal & ((1ull << ln) - 1ull));
      if (ln == 0) Q.wcount[wv] = __popcll(bal);
      __syncthreads();
      base = 0, k0 = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int cw = Q.wcount[w];
        base += (w < wv) ? cw : 0;
        k0 += cw;
      }
      k0 = uni(k0);
    };
    // Rounds.  Round 0 is the block start proper; while enough further rows are violated at the point it reaches
    // (>= BLOCK_MIN_NEW of them: a round costs about as much as that many single-row iterations) another round takes the
    // current working set plus every row violated at the current x -- same independence rule -- and solves for all of them
    // at once.  Every round ends in a valid Goldfarb-Idnani state (x minimises over the working set, multipliers >= 0), so
    // the dual active-set iteration below finishes from wherever the rounds stop.
    // walking: ~1.6 iterations per solve, nothing to gain; 120 variables: two rounds (a third one is worth <1 % there and its
    // third copy of the phase tips the register allocation of the 168-VGPR variant over: 86 spilled registers, 1.43 -> 1.78 ms);
    // three contacts: three (1.13 -> 1.18 M solves/s over two)
    constexpr int BLOCK_ROUNDS = (NC == 3) ? BLOCK_ROUNDS_3C : ((NT >= 256) ? BLOCK_ROUNDS_2C : 1);
    constexpr int CONT_ROUNDS = HMPC_CONT_ROUNDS;  // block rounds of a resumed solve (the continuation variant: 256 VGPRs, a loop fits)
    constexpr bool BLOCK_FRICTION = NT >= 256;
    constexpr int BLOCK_MIN_NEW = (NC == 3) ? BLOCK_MIN_NEW_3C : BLOCK_MIN_NEW_2C;
    constexpr int EPT = (NC == 3 && NT < 512) ? EPT_3C : 5;  // packed-triangle entries per thread during the Schur inversion
    constexpr int KBMAX_3C = (EPT_3C >= 8) ? 63 : ((EPT_3C == 7) ? 59 : 54);
    // Schur matrix of the fast variants on the matrix cores (schur_invert): 3 x 3 tiles = 48 rows for the 120-variable variants,
    // 4 x 4 = 64 rows with three contacts, 5 x 5 = 80 rows for the wide variant (eight waves)
    constexpr int NTGS = (NT >= 512) ? 5 : (NC == 3 ? 4 : (CONT ? 6 : 3));  // (continuation variant: 6 x 6 tiles = its 96 rows)
    // (not the 128-thread variants: measured slower there, profiles/r05/schur_128_ab.txt; not the safe variants: register budget)
    constexpr bool SCHUR_MFMA = !LONGRUN && NT >= 256 && SM::QMAX >= 16 * NTGS;
    constexpr int KBMAX = SCHUR_MFMA ? 16 * NTGS
                                     : ((NT >= 512) ? 71 : ((NT >= 256) ? (NC == 3 ? (KBMAX_3C < SM::QMAX ? KBMAX_3C : SM::QMAX) : 45) : 34));  // KBMAX(KBMAX+1)/2 <= EPT*NT
    static_assert((SCHUR_MFMA || KBMAX * (KBMAX + 1) / 2 <= EPT * NT) && KBMAX <= SM::QMAX, "block start capacity");
    // (one round as a lambda instantiated once per round rather than a loop: with the loop the register allocator keeps
    //  ~50 more VGPRs alive across the whole phase -- measured 185 -> 244 on the unconstrained variants)
    // refresh (safe variants only): the round is there to REBUILD E for the current working set (plus whatever is violated) --
    // no minimum of new rows, and working sets beyond the register-resident inversion's capacity go through the in-place one
    auto block_round = [&](const int round, const bool refresh) __attribute__((always_inline)) -> bool {  // true: another round may follow
    if (round > 0) {
      take = false;
      bool fresh = false;
      if (is_c) {
        const int ac = Q.act[tid];
        bool viol = false;
        if (!refresh && ac == 0 && c_rr <= 6) viol = my_slack(Q.x, side, raw) < -FEAS_TOL;  // (a refresh rebuilds E for the working set AS IT IS)
        if (ac != 0) side = ac;
        const unsigned long long bal = __ballot(viol || ac != 0);
        const bool partner = (c_rr < 4) && ((bal >> (ln ^ 1)) & 1ull);
        fresh = viol && !partner;
        take = (ac != 0) || fresh;
      }
      if constexpr (CONT) {
        // One more independence rule, needed once the Fz cap (row 7) can be in the working set (it enters through single-row
        // iterations only, i.e. never before the fast variants' rounds, but before the continuation variant's): the moment parts
        // of the toe and heel rows 5 and 6 are parallel (+-t1), so their difference is a pure force row, and together with three
        // more pure force rows of the leg-step (one friction row per axis + the cap) the six rows have rank 5.  When the set a
        // round is about to take has that shape in a leg-step, one FRESH row of it stays out: the highest fresh friction row,
        // else the fresh one of rows 6 / 5 (the rows already in the working set are independent by construction).
        const unsigned long long tb = __ballot(take), fb = __ballot(fresh);
        const int sh = ln & ~7;  // rows 8e .. 8e+7 sit in eight consecutive lanes of one wave
        const unsigned t8 = (unsigned)(tb >> sh) & 0xffu, f8 = (unsigned)(fb >> sh) & 0xffu;
        if (__popc(t8 & 0x8fu) == 3 && (t8 & 0x60u) == 0x60u) {
          const unsigned ff = f8 & 0x0fu;
          const int victim = ff ? 31 - __clz((int)ff) : ((f8 & 0x40u) ? 6 : 5);
          if ((tid & 7) == victim && fresh) fresh = false, take = false;
        }
      }
      count_candidates();
      HMPC_DBG(if (k0 > (LONGRUN ? SM::QMAX : KBMAX)) ++dbg_norounds_cap; else if (!refresh && k0 - q < BLOCK_MIN_NEW) ++dbg_norounds_few;)
      if constexpr (SPILLS) {
        if (k0 > KBMAX + HMPC_EARLY_HANDOVER_MARGIN) early_handover = true;
      }
      if (ub((!refresh && k0 - q < BLOCK_MIN_NEW) || k0 > (LONGRUN ? SM::QMAX : KBMAX))) return false;  // not worth a round / does not fit: the iteration below goes on
      // (no barrier here: what follows writes act / slot / Wrow entries that nobody reads before the barrier behind the slot deal,
      //  and wcount is not written again before the release loop, several barriers on)
      ++iters;
      tick = true;  // rows 0..7 may be in the set from here on
    } else
    if (args.wset) {
      if (is_c) {
        int st = (int)S.ls_step[c_e] + args.wset_shift;
        st = st < h ? st : h - 1;
        const int sv = (int)args.wset[(size_t)inst * C8 * h + C8 * st + 8 * (int)S.ls_leg[c_e] + c_rr];
        take = sv != 0;
        side = sv > 0 ? 1 : -1;
      }
      count_candidates();
      tick = k0 > 0;
      if (!tick) __syncthreads();  // wcount is rewritten below
    }
    if (!tick) {  // uniform.  Rows 4-6 violated at the unconstrained minimiser
      take = false;
      if constexpr (BLOCK_FRICTION) {
      // ... and a friction row (0-3) violated there whose partner on the same axis (0<->1, 2<->3) is not: at most one row
      // per axis, so that the rows taken from one leg-step stay linearly independent
      if (is_c && c_rr <= 6) {
        const bool viol = my_slack(Q.xu, side, raw) < -FEAS_TOL;
        const unsigned long long bal = __ballot(viol);
        const bool partner = (c_rr < 4) && ((bal >> (ln ^ 1)) & 1ull);  // rows 8e+rr: the partner sits in the neighbouring lane
        take = viol && !partner;
      }
      } else {
      if (is_c && c_rr >= 4 && c_rr <= 6) take = my_slack(Q.xu, side, raw) < -FEAS_TOL;
      }
      count_candidates();
    }
    const int rlo = (tick || BLOCK_FRICTION) ? 0 : 4, rhi = tick ? 7 : 6;
    bool bad_start = false;
    if constexpr (SPILLS) {
      if (k0 > KBMAX + HMPC_EARLY_HANDOVER_MARGIN) early_handover = true;  // (uniform) far more candidates than this variant can take at once
    }
    if (k0 > (LONGRUN ? SM::QMAX : KBMAX)) k0 = LONGRUN ? SM::QMAX : KBMAX;
    if (is_c) {  // slots are dealt afresh every round (E is rebuilt from scratch)
      const bool in = take && base + below < k0;
      const int sl = in ? base + below : 0;
      Q.act[tid] = in ? (signed char)side : (signed char)0;
      Q.slot[tid] = (unsigned char)sl;
      if (in) Q.Wrow[sl] = (typename SM::Sol::row_t)tid;
    }
    __syncthreads();
    if (k0 > 0) {
      pk_fence();
      // (b) S0(i,j) = n_i' M n_j: rows of leg-steps (e0, e1) meet only in my block
#pragma unroll
      for (int s = 0; s < BPT; ++s)
      if (OWN(s) && E1(s) < ng) {
        const unsigned long long am0 = *reinterpret_cast<const unsigned long long *>(&Q.act[8 * E0(s)]);
        const unsigned long long am1 = *reinterpret_cast<const unsigned long long *>(&Q.act[8 * E1(s)]);
        if (am0 != 0ull && am1 != 0ull) {
          const int leg0 = S.ls_leg[E0(s)], leg1 = S.ls_leg[E1(s)];
          // Every lane walks over ITS OWN active rows (bit scan of the 8 activity bytes of a leg-step).  Counting r1 and r0 through
          // rlo .. rhi instead made each wave execute the union of its lanes' rows -- nearly all 7 x 7 combinations, a 36-FMA block
          // product each, for the 1-2 rows a leg-step really has (blk:S0 14 k cycles per workgroup).  Same arithmetic per entry.
          auto rowmask = [&](const unsigned long long am) __attribute__((always_inline)) -> unsigned {
            const unsigned long long t = am & 0x0101010101010101ull;  // (+1 and -1 both have bit 0 set)
            const unsigned m8 = (unsigned)((t * 0x0102040810204080ull) >> 56);  // bit r = byte r of am is non-zero (no carries: the partial products land on distinct bits)
            return m8 & ((2u << rhi) - 1u) & ~((1u << rlo) - 1u);
          };
          const unsigned m0all = rowmask(am0);
          unsigned m1 = rowmask(am1);
          while (m1 != 0u) {
            const int r1 = __ffs((int)m1) - 1;
            m1 &= m1 - 1u;
            const int ac1 = (int)(signed char)((am1 >> (8 * r1)) & 0xff);
            double cn1[GS], t6[GS];
#pragma unroll
            for (int k = 0; k < GS; ++k) cn1[k] = (double)ac1 * S.Cn[leg1][r1][k];
            blk_rows(s, cn1, t6);  // (a diagonal block is stored as the full symmetric 6x6)
            const int s1 = Q.slot[8 * E1(s) + r1];
            unsigned m0 = DIAG(s) ? (m0all & ((2u << r1) - 1u)) : m0all;  // (diagonal block: r0 <= r1)
            while (m0 != 0u) {
              const int r0 = __ffs((int)m0) - 1;
              m0 &= m0 - 1u;
              const int ac0 = (int)(signed char)((am0 >> (8 * r0)) & 0xff);
              double v = 0.0;
#pragma unroll
              for (int k = 0; k < GS; ++k) v = dfma((double)ac0 * S.Cn[leg0][r0][k], t6[k], v);
              Eat(Q.slot[8 * E0(s) + r0], s1) = v;
            }
          }
        }
      }
      if constexpr (SCHUR_MFMA) {  // (cleared before the barrier that ends the formation of S0: the mat-vec staging is free here)
        if (tid == 0) reinterpret_cast<SchurPanel<NTGS> *>(&Q.ST[0][0])->pn.bad = 0;
      }
      __syncthreads();
      PROF_MARK(P_B_S0);
      // (c) inversion of the k0 x k0 Schur matrix by symmetric sweeps with the packed triangle spread over the threads'
      //     registers (<= EPT entries each); the safe variants also take sets beyond that capacity, swept in place
      //     (LDS, or global for the EGLOBAL variant), one pivot per two barriers -- slow, and only ever run for the handful of
      //     instances per thousand that need hundreds of working-set changes at many times the nominal input ranges
      if constexpr (SCHUR_MFMA) {
        // 4 x 4 block pivots on v_mfma_f64_16x16x4_f64: S0 as 16 x 16 tiles in the accumulators (six tiles on four waves), the
        // pivot panels in the mat-vec staging (free here), E written back over S0 in the packed triangle
        static_assert(sizeof(SchurPanel<NTGS>) <= sizeof(Q.ST), "the Schur panels live in the mat-vec staging");
        SchurPanel<NTGS> &SP = *reinterpret_cast<SchurPanel<NTGS> *>(&Q.ST[0][0]);
        schur_scale_exponents<NTGS>(k0, Ep, SP.kexp);  // (read again at the store, many barriers later; the tile loader takes its own from the diagonal)
        switch (wv) {  // uniform: per-wave specialised code
          case 0: schur_invert<NTGS, NW, 0>(SP, k0, Ep); break;
          case 1: schur_invert<NTGS, NW, 1>(SP, k0, Ep); break;
          case 2: schur_invert<NTGS, NW, (NW > 2 ? 2 : 0)>(SP, k0, Ep); break;  // (cases 2-3: not in the two-wave variants)
          case 3: schur_invert<NTGS, NW, (NW > 2 ? 3 : 0)>(SP, k0, Ep); break;
          case 4: schur_invert<NTGS, NW, (NW > 4 ? 4 : 0)>(SP, k0, Ep); break;  // (cases 4-7: eight-wave variants only)
          case 5: schur_invert<NTGS, NW, (NW > 4 ? 5 : 0)>(SP, k0, Ep); break;
          case 6: schur_invert<NTGS, NW, (NW > 4 ? 6 : 0)>(SP, k0, Ep); break;
          default: schur_invert<NTGS, NW, (NW > 4 ? 7 : 0)>(SP, k0, Ep); break;
        }
        __syncthreads();
        bad_start = SP.pn.bad != 0;
      } else
      if (LONGRUN && (SM::EGLOBAL || ub(k0 > KBMAX))) {  // (EGLOBAL: always in place -- one inversion path less to hold registers for)
        bad_start = false;
        const int ti = tid >> 4, tj = tid & 15;
        for (int sp = 0; sp < k0; ++sp) {
          for (int i = tid; i < k0; i += NT) Q.col[i] = Eat(i, sp);
          __syncthreads();
          const double dv = Q.col[sp];
          bad_start = bad_start || !(dv > 1e-7);
          double idv = __builtin_amdgcn_rcp(dv);
          idv = dfma(dfma(-dv, idv, 1.0), idv, idv);
          idv = dfma(dfma(-dv, idv, 1.0), idv, idv);
          for (int ib = 0; ib < k0; ib += NT / 16) {
            const int i = ib + ti;
            if (i < k0) {
              const double ci = Q.col[i] * idv;
              for (int j = tj; j <= i; j += 16) {
                double &ee = Ep[(unsigned)(i * (i + 1) / 2 + j)];
                const double cj = Q.col[j];
                ee = (i == sp) ? ((j == sp) ? -idv : cj * idv) : ((j == sp) ? ci : dfma(-ci, cj, ee));
              }
            }
          }
          __syncthreads();
        }
        for (int t = tid; t < k0 * (k0 + 1) / 2; t += NT) Ep[(unsigned)(t)] = -Ep[(unsigned)(t)];  // the sweeps leave -S0^-1
        __syncthreads();
      } else {
        const int npair = k0 * (k0 + 1) / 2;
        const int nept = (npair + NT - 1) / NT;
        bad_start = false;
        double er[EPT];
        int ei[EPT], ej[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          const int t = tid + NT * u;
          int i = 0, j = 0;
          double v = 0.0;
          if (t < npair) {
            i = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            while (i * (i + 1) / 2 > t) --i;
            j = t - i * (i + 1) / 2;
            v = Ep[(unsigned)(t)];
          } else {
            i = -1, j = -1;
          }
          er[u] = v, ei[u] = i, ej[u] = j;
        }
        // Two pivots per barrier: K = {s, s+1}, D = A(K,K), G = D^-1 in closed form (one reciprocal),
        //     A(K,K) <- -G,   A(i,K) <- P_i G,   A(i,j) <- A(i,j) - P_i G P_j'      with P_i = [A(i,s), A(i,s+1)].
        // The two pivot columns of the next step are published right after the update (two pairs of LDS vectors,
        // alternating); an odd last pivot takes one scalar sweep.
        auto publish = [&](double *buf, int c, int i, int j, double v) __attribute__((always_inline)) {
          if (j == c) buf[i] = v;
          else if (i == c) buf[j] = v;
        };
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          if (ei[u] >= 0) {
            publish(Q.col, 0, ei[u], ej[u], er[u]);
            publish(Q.z, 1, ei[u], ej[u], er[u]);
          }
        }
        __syncthreads();
        int s = 0, par = 0;
        for (; s + 1 < k0; s += 2, par ^= 1) {
          const double *ca = par ? Q.w : Q.col, *cb = par ? Q.d : Q.z;
          double *na = par ? Q.col : Q.w, *nb = par ? Q.z : Q.d;
          const double a00 = ca[s], a01 = ca[s + 1], a11 = cb[s + 1];
          const double det = dfma(a00, a11, -(a01 * a01));
          // both pivots of a positive definite Schur matrix are >= ~2e-3 |n|^2 (the second one is det / a00)
          bad_start = bad_start || !(a00 > 1e-7) || !(det > 1e-7 * a00);
          double idet = __builtin_amdgcn_rcp(det);
          idet = dfma(dfma(-det, idet, 1.0), idet, idet);
          idet = dfma(dfma(-det, idet, 1.0), idet, idet);
          const double g00 = a11 * idet, g01 = -a01 * idet, g11 = a00 * idet;
#pragma unroll
          for (int u = 0; u < EPT; ++u) {
            if (u < nept) {  // uniform: only the register slots this k0 actually uses
              const int i = ei[u], j = ej[u];
              const int ic = i < 0 ? 0 : i, jc = j < 0 ? 0 : j;
              const double pi0 = ca[ic], pi1 = cb[ic], pj0 = ca[jc], pj1 = cb[jc];
              const double ti0 = dfma(pi0, g00, pi1 * g01), ti1 = dfma(pi0, g01, pi1 * g11);
              const double tj0 = dfma(pj0, g00, pj1 * g01), tj1 = dfma(pj0, g01, pj1 * g11);
              const double upd = dfma(-ti1, pj1, dfma(-ti0, pj0, er[u]));
              const bool iK = (i == s) || (i == s + 1), jK = (j == s) || (j == s + 1);
              const double vkk = (i == s) ? -g00 : ((j == s) ? -g01 : -g11);  // i >= j inside K
              er[u] = iK ? (jK ? vkk : ((i == s) ? tj0 : tj1)) : (jK ? ((j == s) ? ti0 : ti1) : upd);
              publish(na, s + 2, i, j, er[u]);
              publish(nb, s + 3, i, j, er[u]);
            }
          }
          __syncthreads();
        }
        if (s < k0) {
          const double *cs = par ? Q.w : Q.col;
          const double dv = cs[s];
          bad_start = bad_start || !(dv > 1e-7);
          double idv = __builtin_amdgcn_rcp(dv);
          idv = dfma(dfma(-dv, idv, 1.0), idv, idv);
          idv = dfma(dfma(-dv, idv, 1.0), idv, idv);
#pragma unroll
          for (int u = 0; u < EPT; ++u) {
            if (u < nept) {
              const int i = ei[u], j = ej[u];
              const double ci = cs[i < 0 ? 0 : i] * idv, cj = cs[j < 0 ? 0 : j];
              const double upd = dfma(-ci, cj, er[u]);
              er[u] = (i == s) ? ((j == s) ? -idv : cj * idv) : ((j == s) ? ci : upd);
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u)
          if (ei[u] >= 0) Ep[(unsigned)(tid + NT * u)] = -er[u];  // the sweeps leave -S0^-1
      }
      q = k0;
      PROF_MARK(P_B_INV);
      HMPC_DBG(++dbg_rounds; if (bad_start) ++dbg_bad;)
      if (ub(bad_start)) {
        // only possible for a working set inherited from the previous tick whose rows have become (nearly) dependent
        // under this tick's data: forget it and start from the empty set
        for (int t = tid; t < SM::MMAX; t += NT) Q.act[t] = 0, Q.slot[t] = 0;
        if (is_v) Q.x[tid] = Q.xu[tid];
        q = 0;
        __syncthreads();
      }
      if (q > 0) {
      // (d) multipliers u = E (b - N x_u); rows with a negative multiplier do not belong to the working set: remove the
      //     most negative one, update E by the Schur complement, repeat
      active_residual(Q.xu);
      __syncthreads();
      {
        double dmy = INF;
        int dj = 0;
        e_times(Q.d, Q.u, false, dmy, dj, false);
      }
      __syncthreads();
      // Rows of the foot-x moment window (row 4: 0 <= t0.M <= 0.01, both bounds finite and a hair apart) are active at
      // almost every stance leg-step of the optimum, but x_u says on which SIDE only half reliably: a negative multiplier
      // there means "the other bound", not "inactive".  Such rows are switched to their other bound -- all of them at once,
      // each row at most once -- instead of being released and added again later: n_j -> -n_j, b_j -> the other bound, so
      // E(a,b) -> s_a s_b E(a,b) (no new inversion), then u = E (b - N x_u) again.  Measured on the 2-contact set: 8.9 ->
      // about 4.4 working-set changes per solve (offline emulation; GPU: mean iterations 9.1 -> see profiles/r02).  Same optimum (the QP is strictly convex; the final KKT check is unchanged).
      // (not in a refresh: there the working set is a Goldfarb-Idnani state already -- jumping to another dual-feasible set could
      //  lower the dual objective and make the iteration cycle; multipliers that come out negative by round-off are left to the
      //  ratio test of the next step)
      while (!refresh) {
        double um = (tid < q) ? Q.u[tid] : INF;
        bool cand = false;
        if (tid < q && um < -1e-12) {
          const int c = Q.Wrow[tid];
          cand = ((c & 7) == 4) && (ANTICYCLE ? (Q.flpc[c] & 1) == 0 : Q.flpc[c] == 0);
        }
        const double wmin = wave_min(um);
        const unsigned long long b2 = __ballot(um == wmin);
        const int wl = (int)__ffsll((long long)b2) - 1;
        // (the decision below must come from data no thread modifies before every thread has read it: the winning lane of
        // each wave publishes whether it is itself such a row)
        if (ln == wl) Q.redv[wv] = um, Q.redi[wv] = tid, Q.wcount[wv] = cand ? 1 : 0;
        __syncthreads();
        int l = Q.redi[0], lcand = Q.wcount[0];
        double umin = Q.redv[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const double ov = Q.redv[w];
          if (ov < umin) umin = ov, l = Q.redi[w], lcand = Q.wcount[w];
        }
        l = uni(l);
        if (ub(!(umin < -1e-12))) break;
        ++iters;
        // ... but only when the most negative multiplier is itself such a row: when it belongs to a toe/heel row that should
        // not be in the set, that row goes first (its presence is what turns a whole group of window multipliers negative;
        // switching all of them at once then costs a release and a re-entry each: p99 of the working-set changes 50 instead of 19)
        if (ub(lcand != 0)) {
          if (tid < q) Q.r[tid] = cand ? -1.0 : 1.0;
          if (cand) {
            const int c = Q.Wrow[tid];
            Q.act[c] = (signed char)(-Q.act[c]);
            Q.flpc[c] = ANTICYCLE ? (unsigned char)(Q.flpc[c] | 1) : (unsigned char)1;
          }
          __syncthreads();
          {
            const int ti = tid >> 4, tj = tid & 15;
            for (int ib = 0; ib < q; ib += NT / 16) {
              const int i = ib + ti;
              if (i < q) {
                const double si = Q.r[i];
                for (int j = tj; j < i; j += 16) Ep[(unsigned)(i * (i + 1) / 2 + j)] *= si * Q.r[j];
              }
            }
          }
          active_residual(Q.xu);
          __syncthreads();
          {
            double dmy = INF;
            int dj = 0;
            e_times(Q.d, Q.u, false, dmy, dj, false);
          }
          __syncthreads();
          continue;
        }
        drop_slot(l, true, Q.u[l]);  // the multipliers follow the downdate: no second product with E
        if (q == 0) break;
      }
      PROF_MARK(P_B_DROP);
      // (e) x = x_u + M N' u
      if (q > 0) {
        gather_w(Q.u, 1.0, 0.0);
        __syncthreads();
        rmatvec(Q.w);
        if (is_v) Q.x[tid] = Q.xu[tid] + Q.z[tid];
        __syncthreads();
      } else {
        if (is_v) Q.x[tid] = Q.xu[tid];  // every row was released again
        __syncthreads();
      }
      }
    }
    return !ub(k0 == 0 || q == 0);  // (nothing to build on otherwise)
    };  // block_round
  if constexpr (CONT) {
    // a resumed solve: the state handed over is a Goldfarb-Idnani state, so further ROUNDS apply as they stand -- the working set
    // plus every row violated at the point reached, all at once, up to this variant's capacity
    if (resumed) {
      // (instantiated one by one: written as a loop the phase keeps ~100 more registers alive -- 104 spilled at the 256 this variant has)
      bool more = true;
      if constexpr (CONT_ROUNDS > 0) more = block_round(1, false);
      if constexpr (CONT_ROUNDS > 1) {
        if (more) more = block_round(2, false);
      }
      if constexpr (CONT_ROUNDS > 2) {
        if (more) more = block_round(3, false);
      }
      if constexpr (CONT_ROUNDS > 3) {
        if (more) more = block_round(4, false);
      }
      static_assert(CONT_ROUNDS >= 0 && CONT_ROUNDS <= 4, "rounds are instantiated one by one");
    }
  }
  // (the safe variants as well since round 6: their first pass over a flagged instance starts from the block start like any other
  //  solve -- 40-60 iterations instead of the 150-250 of a cold run at 6x the input ranges, on ONE workgroup at the tail of the
  //  stream --; the last-resort passes with perturbed bounds stay cold)
  if (args.warm && !(RESUMABLE && resumed)) {
    bool more = block_round(0, false);
    if constexpr (BLOCK_ROUNDS > 1) {
      if (more) more = block_round(1, false);
    }
    if constexpr (BLOCK_ROUNDS > 2) {
      if (more) more = block_round(2, false);
    }
    static_assert(BLOCK_ROUNDS <= 3, "rounds are instantiated one by one");
  }
  PROF_MARK(P_BLOCK);

  // =============================== Q: dual active set (Goldfarb-Idnani, range-space form) ===============================
  // Safe-pass variants (LONGRUN): E = (N_W M N_W')^-1 is kept current by rank-one updates (bordering / Schur downdates), whose round-off adds up
  // over the hundreds of working-set changes a run at many times the nominal input ranges takes (measured: a 470-iteration
  // run ended 3.5 N off its constraints while every multiplier looked fine).  So every REFRESH_EVERY changes -- and once more
  // before the final refinement when enough have happened since -- E is REBUILT from the register blocks of M for the
  // current working set (a block round in refresh mode: S0 formed block-locally, inverted from scratch, multipliers and
  // point recomputed, rows whose multiplier comes out negative released).  The fast variants never run long enough to need it.
  constexpr int REFRESH_EVERY = 48, REFRESH_FINAL = 12;
  // (the continuation variant runs the hard instances -- dozens to hundreds of further working-set changes on top of the fast
  //  variant's -- and rebuilds E periodically as well: on the matrix cores, a refresh costs about four single-row iterations)
  constexpr bool REFRESHES = LONGRUN;  // (the continuation variant: measured no difference at 6x, and a seventh instantiation of the round costs it 113 spilled registers)
  int since_refresh = 0;
  for (int pass = 0; pass < (ANTICYCLE ? 8 : 3) && code == S_OK; ++pass) {
    const int iters_at_entry = uni(iters);  // (uniform: a scalar register)
    if constexpr (ANTICYCLE) perturbed = false;
    // ---- main loop ----
    while (true) {
      if constexpr (REFRESHES) {
        if (ub(since_refresh >= REFRESH_EVERY && q > 0)) {
          __syncthreads();
          (void)block_round(3, true);
          since_refresh = 0;
        }
      }
      // (1) most violated constraint; the winning lane of each wave also publishes its constants
      double val = INF, raw = INF;
      int side = 1;
      if constexpr (ANTICYCLE) {
        if (is_c && !c_ignored && Q.act[tid] == 0 && (Q.flpc[tid] >> 1) < HMPC_READD_LIMIT) val = my_slack(Q.x, side, raw);
      } else {
        if (is_c && !c_ignored && Q.act[tid] == 0) val = my_slack(Q.x, side, raw);
      }
      PROF_MARK(P_T1);
      {
        const double wmin = wave_min(val);
        const unsigned long long bal = __ballot(val == wmin);
        const int wl = (int)__ffsll((long long)bal) - 1;
        if (ln == wl) {
          auto &rc = Q.rec[wv];
          rc.val = val;
          rc.raw = raw;
          rc.idx = tid;
          rc.side = side;
          const double *c_cn = row_cn();
#pragma unroll
          for (int k = 0; k < 6; ++k) rc.cn[k] = c_cn[k];
        }
      }
      PROF_MARK(P_SEL_A);
      __syncthreads();
      int wsel = 0;
      double pval = Q.rec[0].val;
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const double ov = Q.rec[w].val;
        if (ov < pval) pval = ov, wsel = w;
      }
      PROF_MARK(P_SEL);
      if (ub(!(pval < -FEAS_TOL))) break;
      if constexpr (ANTICYCLE) {
        if (ub(pval < -1e8)) {  // the iterate has left the problem's scale (an ill-conditioned working set blew E up): flagged at once
          code = S_KKT;
          break;
        }
      }
      if constexpr (ANTICYCLE) {
        // Degenerate vertices, handled where they occur (round 6; until then only the host's last-resort passes did this, after a
        // cold re-solve had burnt its whole iteration bound).  When the re-addition counter of a row reaches its limit, or a
        // resumed solve's budget runs out, the instance is cycling: every bound is moved outward by relax (1 + frac(0.618 row)) --
        // a different amount per row, which separates the coinciding vertices --, the multipliers and the point are moved onto
        // the perturbed problem's vertex by the refinement step below (u += E (b' - N x), x = x_u + M N'u: x(u) is linear in u),
        // and the iteration goes on in the next pass.  Up to three levels, 1e-7 / 1e-6 / 1e-5.  The epilogue re-solves on the final
        // working set with the EXACT bounds and repeats the exact KKT check: passed = HMPC_S_OK, exact.
        // (safe variants only.  The continuation variant's long-runners were measured to run through every level's budget and end
        //  flagged all the same -- 512 instead of 128 iterations at the tail of its launch --: there the budget and the KKT check decide,
        //  and the safe pass, which starts cold, perturbs)
        const bool cycling = LONGRUN && ub(S.pad0 != 0);
        if (cycling) {
          const double rl = uni_d(S.relax);
          if (rl < 0.99e-5) {  // (uniform) another level left
            __syncthreads();  // (everyone has read pad0 and relax)
            if (tid == 0) S.relax = (rl == 0.0) ? 1e-7 : 10.0 * rl, S.pad0 = 0;
            for (int t = tid; t < SM::MMAX; t += NT) Q.flpc[t] = (unsigned char)(Q.flpc[t] & 1);
            __syncthreads();
            if constexpr (!LAZY) {
              if (is_c) c_ub_r = row_ub_calc();
            }
            if constexpr (CONT) {  // (a resumed solve's budget starts afresh)
              if (budgeted) itmax = (iters + HMPC_CONT_ITER_BUDGET < itmax_full) ? iters + HMPC_CONT_ITER_BUDGET : itmax_full;
            }
            perturbed = true;
            break;  // -> the refinement moves (x, u) onto the perturbed vertex, the next pass goes on from there
          }
          if (tid == 0) S.pad0 = 0;  // (no level left: the rows that hit the limit stay set aside, the KKT check decides)
        }
      }
      if (iters >= itmax) {
        if constexpr (CONT) {
          if (budgeted) {  // (uniform)
            budget_hit = true;
            HMPC_DBG(dbg_viol = -pval;)
            break;
          }
        }
        code = S_MAXITER;
        break;
      }
      if constexpr (SPILLS) {
        // a violated row and a full working set: stop HERE, between two iterations, where (x, u, W, E) is a complete
        // Goldfarb-Idnani state that the continuation variant can take over (inside an iteration -- after partial steps for
        // the row being added -- it is not).  Conservative by at most one row: the step might have dropped a row first.
        if (q >= SM::QMAX || (early_handover && args.spill && inst < args.spill_cap)) {
          code = S_WORKSET;
          break;
        }
      }
      wsel = uni(wsel);
      const int p = uni(Q.rec[wsel].idx), sgi = uni(Q.rec[wsel].side);
      const int ep = p >> 3;
      double sp = uni_d(Q.rec[wsel].raw);  // (uniform values of the iteration live in scalar registers: sp, up, delta, gamma, t1)
      const double sg = (double)sgi;
      double np[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) np[k] = uni_d(sg * Q.rec[wsel].cn[k]);  // the same value in every lane: scalar registers
      double up = 0.0;
      bool added = false;
      while (!added) {
        ++iters;
        if (iters > itmax) {
          code = S_MAXITER;
          break;
        }
        pk_fence();
        // (2) d_j = n_j' M n+ for the active rows: the rows of leg-step eo meet n+ (on leg-step ep) only in block
        //     (min(eo,ep), max(eo,ep)), whose owner forms t = M(eo,ep) n+ once and dots it with each active row;
        //     the diagonal block also gives gamma = n+' M n+
#pragma unroll
        for (int s = 0; s < BPT; ++s)
        if (OWN(s) && E1(s) < ng && (E0(s) == ep || E1(s) == ep)) {
          // t = M(eo, ep) n+: the row product when ep is this block's column leg-step (and on the diagonal), the column
          // product otherwise; both are formed (no divergence inside the wave) and one is kept
          double t6[GS], tc[GS];
          blk_rows(s, np, t6);
          blk_cols(s, np, tc);
          const bool use_rows = (E1(s) == ep);
#pragma unroll
          for (int k = 0; k < GS; ++k) t6[k] = use_rows ? t6[k] : tc[k];
          const int eo = (E0(s) == ep) ? E1(s) : E0(s);
          const int lego = S.ls_leg[eo];
          const unsigned long long am = *reinterpret_cast<const unsigned long long *>(&Q.act[8 * eo]);
          const unsigned long long sm = *reinterpret_cast<const unsigned long long *>(&Q.slot[8 * eo]);
          if (am != 0ull) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int ac = (int)(signed char)((am >> (8 * rr)) & 0xff);
              if (ac != 0) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < GS; ++k) v = dfma(S.Cn[lego][rr][k], t6[k], v);
                Q.d[(int)((sm >> (8 * rr)) & 0xff)] = (double)ac * v;
              }
            }
          }
          if (DIAG(s)) {
            double gm = 0.0;
#pragma unroll
            for (int k = 0; k < GS; ++k) gm = dfma(np[k], t6[k], gm);
            Q.gamma = gm;
          }
        }
        __syncthreads();
        PROF_MARK(P_D);
        // (3) r = E d; the lane that owns r_j also bids for the dual step length t1 = min_j u_j / r_j over r_j > 0
        {
          double t1c = INF;
          int t1j = 0;
          e_times(Q.d, Q.r, false, t1c, t1j, true);
          const double wmin = wave_min(t1c);
          const unsigned long long bal = __ballot(t1c == wmin);
          const int wl = (int)__ffsll((long long)bal) - 1;
          if (ln == wl) {
            Q.redv[wv] = t1c;
            Q.redi[wv] = t1j;
          }
        }
        __syncthreads();
        PROF_MARK(P_ED);
        // (4) w = n+ - N_W' r
        gather_w(Q.r, -1.0, (is_v && v_e == ep) ? np[v_k] : 0.0);
        __syncthreads();
        PROF_MARK(P_W);
        // (5) z = M w
        rmatvec(Q.w);
        PROF_MARK(P_MV);
        // (6) step lengths and the step
        double delta = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) delta = dfma(np[k], Q.z[GS * ep + k], delta);
        delta = uni_d(delta);
        const double gamma = uni_d(Q.gamma);
        int l = Q.redi[0];
        double t1 = Q.redv[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const double ov = Q.redv[w];
          if (ov < t1) t1 = ov, l = Q.redi[w];
        }
        l = uni(l);
        t1 = uni_d(t1);
        // n+' (M - M N'E N M) n+ against n+' M n+: the part of the new row's normal that is NOT in the span of the working set's
        // (1e-10 and 1e-8 measured on the continuation variant at 6x the input ranges: 20 -> 18 unfinished of 2 081, no difference in time)
        const bool dep = ub(!(delta > 1e-12 * gamma));
        const double t2 = dep ? INF : -sp / delta;
        const double t = (t1 < t2) ? t1 : t2;
        if (ub(t == INF)) {
          // Row p is (numerically) a combination of the working set and no multiplier can absorb it.  F = M = 0 satisfies
          // every row of this QP, so the problem is never infeasible: the event only occurs at degenerate vertices, for a
          // redundant row violated by accumulated round-off.  The row is set aside and the iteration goes on; the final
          // KKT check (which looks at every row again, relative to the force scale) decides the status.
          if (tid == p) c_ignored = true;
          HMPC_DBG(++dbg_dep;)
          break;
        }
        if (!dep && is_v) Q.x[tid] = dfma(t, Q.z[tid], Q.x[tid]);
        if (tid < q) Q.u[tid] = dfma(-t, Q.r[tid], Q.u[tid]);
        up = uni_d(up + t);
        if (!dep) sp = uni_d(dfma(t, delta, sp));
        const bool fullstep = ub(!dep && !(t1 < t2));
        if (fullstep && q >= SM::QMAX) {
          code = S_WORKSET;
          break;
        }
        if (fullstep) {
          // full step: constraint p joins the working set; bordered update of E (16-wide thread tiles of the lower triangle)
          const double idl = 1.0 / delta;
          const int ti = tid >> 4, tj = tid & 15;
          for (int ib = 0; ib < q; ib += NT / 16) {
            const int i = ib + ti;
            if (i < q) {
              const double ri = Q.r[i] * idl;
              for (int j = tj; j <= i; j += 16) {
                double &ee = Ep[(unsigned)(i * (i + 1) / 2 + j)];
                ee = dfma(ri, Q.r[j], ee);
              }
            }
          }
          if (tid < q) Ep[(unsigned)(q * (q + 1) / 2 + tid)] = -Q.r[tid] * idl;
          if (tid == q) {
            Ep[(unsigned)(q * (q + 1) / 2 + q)] = idl;
            Q.u[q] = up;
            Q.Wrow[q] = (typename SM::Sol::row_t)p;
            Q.act[p] = (signed char)sgi;
            Q.slot[p] = (unsigned char)q;
            if constexpr (ANTICYCLE) {
              const unsigned fc = Q.flpc[p];
              if ((fc >> 1) < 127u) Q.flpc[p] = (unsigned char)(fc + 2u);
              if ((fc >> 1) + 1u >= (unsigned)HMPC_READD_LIMIT) S.pad0 = 1;  // this row keeps coming back: the loop head reacts
            }
          }
          ++q;
          added = true;
          __syncthreads();
        } else {
          drop_slot(l);  // partial (or pure dual) step: slot l leaves
        }
        if constexpr (REFRESHES) ++since_refresh;
        PROF_MARK(P_UPD);
      }
      if (code != S_OK) break;
    }
    if (code != S_OK || q == 0) break;
    // nothing happened in this pass: either the refined point of the previous pass is feasible, or (first pass) the block
    // start already is the optimum -- its x, u, E come straight from the inversion, there is nothing to refine
    if (iters == iters_at_entry && !(CONT && budget_hit) && !(ANTICYCLE && perturbed)) break;
    if constexpr (LONGRUN) {
      if (ub(since_refresh >= REFRESH_FINAL)) {  // the answer is read off a freshly built E; the loop above then confirms it
        __syncthreads();
        (void)block_round(3, true);
        since_refresh = 0;
        continue;
      }
    }

    // ---- refinement of the multipliers on the final working set: u += E (b_W - N_W x(u)), x(u) = x_u + M N_W' u ----
    for (int it = 0; it <= HMPC_REFINE; ++it) {
      // (the first correction takes its residual at the iterate the loop above left -- x was moved along with u
      //  step by step, and what the correction is there to remove is the drift of E's rank-one updates, orders of magnitude above
      //  the difference between that x and x(u) -- instead of recomputing x(u) first: one gather + one product with M less)
      if (!(!LONGRUN && it == 0 && HMPC_REFINE > 0)) {
      gather_w(Q.u, 1.0, 0.0);
      __syncthreads();
      rmatvec(Q.w);
      if (is_v) Q.x[tid] = Q.xu[tid] + Q.z[tid];
      __syncthreads();
      }
      if (it == HMPC_REFINE) break;
      active_residual(Q.x);
      __syncthreads();
      double dmy = INF;
      int dj = 0;
      e_times(Q.d, Q.u, true, dmy, dj, false);
      __syncthreads();
    }
    PROF_MARK(P_POLISH);
    if constexpr (CONT) {
      if (budget_hit) break;  // (no further pass: the KKT check below decides)
    }
    // a refinement that moved x across another constraint sends us back into the main loop (rare)
  }

  if constexpr (SPILLS) {
    if (args.spill_slot) {  // uniform
      int slot = -1;
      if (code == S_WORKSET && args.spill) {  // uniform: hand the live state over (SpillLayout)
        __syncthreads();
        // instance i owns slot i (no counter to reset, no atomics, the same place every run); instances beyond the buffer's
        // capacity -- batches above the host's slot cap -- are flagged as before and re-solved cold
        if (inst < args.spill_cap) {
          slot = inst;
          double *sp = reinterpret_cast<double *>(args.spill + (size_t)slot * args.spill_stride);
          const double *qa = reinterpret_cast<const double *>(&Q);
          const double *qb = reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(&Q) + SPL::B_OFF);
          for (int t = tid; t < (int)(SPL::A_BYTES / 8); t += NT) sp[t] = qa[t];
          for (int t = tid; t < (int)(SPL::B_BYTES / 8); t += NT) sp[SPL::A_BYTES / 8 + t] = qb[t];
          for (int t = tid; t < q * (q + 1) / 2; t += NT) sp[SPL::E_OFF / 8 + t] = Ep[(unsigned)t];
          double *mb = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(sp) + args.spill_stride - SPL::M_BYTES);
#pragma unroll
          for (int ii = 0; ii < GS; ++ii)
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) mb[(ii * GS + jj) * NT + tid] = a[0][ii][jj];
        }
      }
      if (tid == 0) args.spill_slot[inst] = slot;
    }
  }
  PROF_MARK(P_POLISH);
  // final KKT check: primal slack (every row not in the working set), residual of the working set's rows, multiplier signs
  __syncthreads();
  auto kkt_ok = [&]() __attribute__((always_inline)) -> bool {
    double val = INF, raw;
    int side;
    if (is_c) {
      const int ac = Q.act[tid];
      if (ac == 0) {
        val = my_slack(Q.x, side, raw);  // set-aside rows are checked again here
      } else {
        // a row of the working set must sit ON its bound: x comes from x_u + M N' u with u from the explicitly updated Schur
        // inverse E, and hundreds of rank-one updates of E (safe-variant runs at many times the nominal input ranges) can
        // drift until N_W x = b_W no longer holds -- that must end as HMPC_S_KKT, never as a quietly wrong "ok"
        double sx = 0.0;
        const double *xp = Q.x + GS * c_e;
        const double *c_cn = row_cn();
#pragma unroll
        for (int k = 0; k < 6; ++k) sx = dfma(c_cn[k], xp[k], sx);
        const double bnd = (ac > 0) ? row_lo() : row_ub();
        const double sc = ((tid & 7) == 7 && ac < 0) ? S.sc7[c_e] : 1.0;  // (the Fz cap on a unit scale, as in my_slack)
        val = -__builtin_fabs(sx - bnd) * sc;
      }
    }
    double umin = (tid < q) ? Q.u[tid] : INF;
    val = wave_min(val);
    umin = wave_min(umin);
    const double xm = wave_min(is_v ? -__builtin_fabs(Q.x[tid]) : 0.0);
    if (ln == 0) Q.redv[wv] = val, Q.redw[wv] = umin, Q.rec[wv].raw = xm;  // (the three reductions share one barrier; the selection
                                                                        //  records are free here -- and not a byte of LDS is added:
                                                                        //  the 128-thread variants fit six workgroups per CU by 300 B)
    __syncthreads();
    double xmax = 1.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      val = (Q.redv[w] < val) ? Q.redv[w] : val;
      umin = (Q.redw[w] < umin) ? Q.redw[w] : umin;
      xmax = (-Q.rec[w].raw > xmax) ? -Q.rec[w].raw : xmax;
    }
    // (no barrier behind the reads: the only other writer of the three arrays -- a second call in the last-resort pass -- comes
    //  after the barriers of that pass)
    constexpr double KKT_TOL = LONGRUN ? 2e-5 : 2e-6;
    return !(val < -KKT_TOL * xmax || umin < -1e-6 * xmax);  // relative to the force scale
  };
  if (code == S_OK && !ub(kkt_ok())) code = S_KKT;  // (the reductions leave the same values in every lane: a scalar decision)
  if ((ANTICYCLE ? ub(S.relax != 0.0) : ub(args.relax != 0.0)) && code == S_OK) {
    // Last-resort pass (bounds moved outward, hmpc_resolve_failed): the perturbation was only there to separate coinciding
    // vertices.  With the working set it ended on, the multipliers and the point are re-solved for the EXACT bounds
    // (x(u) is linear in u: one correction u += E (b_W - N_W x(u)) lands on the exact vertex) and the KKT check is repeated
    // with the exact bounds.  Passed: the instance is solved exactly, HMPC_S_OK.  Not passed (the perturbed problem's
    // working set is not optimal for the exact one): the perturbed answer is kept and reported as HMPC_S_OK_RELAXED.
    __syncthreads();
    const double relax_was = uni_d(S.relax);  // (uniform: scalar registers)
    if (is_v) Q.col[tid] = Q.x[tid];
    if (tid < q) Q.r[tid] = Q.u[tid];  // (Q.r is free here: the multipliers of the perturbed problem, restored with x below)
    __syncthreads();
    if (tid == 0) S.relax = 0.0;
    __syncthreads();
    if constexpr (!LAZY) {
      if (is_c) c_ub_r = row_ub_calc();
    }
    active_residual(Q.x);
    __syncthreads();
    {
      double dmy = INF;
      int dj = 0;
      e_times(Q.d, Q.u, true, dmy, dj, false);
    }
    __syncthreads();
    gather_w(Q.u, 1.0, 0.0);
    __syncthreads();
    rmatvec(Q.w);
    if (is_v) Q.x[tid] = Q.xu[tid] + Q.z[tid];
    __syncthreads();
    if (!ub(kkt_ok())) {
      // the perturbed problem's answer is reported as it was: point, multipliers AND bounds (the objective below is formed from
      // u and the bounds: exact-bound multipliers next to the restored x would describe neither problem)
      __syncthreads();
      if (is_v) Q.x[tid] = Q.col[tid];
      if (tid < q) Q.u[tid] = Q.r[tid];
      if (tid == 0) S.relax = relax_was;
      code = S_OK_RELAXED;
      __syncthreads();
    }
  }

  // ---------------- output: forces, working set, status word, flag lists, objective (stage_output above)
  {
    // (an instance that ran into the CALLER'S iteration cap is the caller's answer: it is neither counted nor listed for the
    //  safe pass -- the device-side repair would otherwise re-solve it cold and overwrite its last iterate)
    const bool capped = (code == S_MAXITER) && (args.iter_cap > 0 && args.iter_cap < itmax_v);
    if constexpr (REGULARISES) code = (args.reg_step == 1 && code == S_OK) ? (int)S_REG_STEP : code;  // x_1 of the regularised QP: one more step to go
    stage_output<NMAX, HMAX, NT, QCAP, NC, BPT>(S, args, inst, h, n, q, iters, code, capped, RESUMABLE && resumed);
#ifdef HMPC_DEBUG_STATS
    if (tid == 0 && args.obj64) args.obj64[inst] = (dbg_viol > 0.0) ? -dbg_viol : (double)(dbg_bad + 10 * dbg_rounds + 1000 * dbg_norounds_cap + 10000 * dbg_norounds_few + 100000 * dbg_dep);
#endif
  }
  PROF_MARK(P_FINAL);
  PROF_FLUSH();
}

}  // namespace hmpc
