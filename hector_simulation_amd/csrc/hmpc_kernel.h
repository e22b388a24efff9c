// hmpc_kernel.h -- the fused assembly + QP-solve kernel (one 256-thread workgroup per MPC instance, gfx950).
//
// Replaces, for a whole batch at once, the reference's per-tick CPU path
//   update_problem_data -> solve_mpc -> qpOASES::QProblem::init
//   (ConvexMPC/convexMPC_interface.cpp:83-103, ConvexMPC/SolverMPC.cpp:371-738, third_party/qpOASES/src/QProblem.cpp:316).
//
// Phases (all state lives in LDS / registers; HBM sees only the ~716 B record in and 12h floats + 1 word out):
//   A  assembly in binary32 under the HMPC-A1 arithmetic contract (bit-identical to oracle/hmpc_oracle.c):
//      trig -> scalar algebra -> Acd^k, Phi_k = Acd^k Bcd -> tracking error -> swing elimination tables
//      -> H = 2(B'SB + alpha) on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 = k-ordered fmaf chain), g.
//   S  M = H^-1 in binary64 by n symmetric sweeps, matrix held in registers (8x4 tile per thread), one barrier a sweep.
//   Q  dual active-set (Goldfarb-Idnani in range-space form): Schur inverse E = (N M N')^-1 kept explicitly and
//      updated by bordering / Schur-complement rank-1 steps, so every iteration is parallel mat-vecs -- no
//      triangular solves.  Two steps of iterative refinement of the multipliers at the end.
// LDS layout: one (n x (n+1)) binary64 square holds M in its upper triangle (diagonal included) and E packed in
// its strict lower triangle (E(i,j), i>=j, at row i+1, column j); assembly scratch and solver vectors share a union.
#pragma once
#include <stdint.h>

#include "hmpc_math.h"

namespace hmpc {

constexpr int NT = 256;  // threads per workgroup
constexpr int NW = NT / 64;

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
  long long *prof;  // optional [batch][HMPC_NPROF] per-phase shader-clock cycles (thread 0's view), profiling builds only
};
constexpr int NPROF = 24;
enum : int { P_ASM = 0, P_HG, P_SWEEP, P_XU, P_SEL, P_D, P_ED, P_W, P_MV, P_T1, P_UPD, P_POLISH, P_FINAL, P_TOTAL, P_SW_RD, P_SW_FMA, P_SW_PUB, P_SW_BAR };
#ifdef HMPC_PROFILE
#define PROF_DECL long long _pt = clock64(), _pt0 = _pt; long long _pacc[NPROF] = {0}
#define PROF_MARK(ph) do { long long _n = clock64(); _pacc[ph] += _n - _pt; _pt = _n; } while (0)
#define PROF_FLUSH() do { if (threadIdx.x == 0 && args.prof) { _pacc[P_TOTAL] = clock64() - _pt0; for (int _i = 0; _i < NPROF; ++_i) args.prof[(size_t)inst * NPROF + _i] = _pacc[_i]; } } while (0)
#else
#define PROF_DECL
#define PROF_MARK(ph)
#define PROF_FLUSH()
#endif

// offsets (in floats) of the debug dump, shared with the host
template <int NMAX>
struct DbgLayout {
  static constexpr int H = 0, G = NMAX * NMAX, FC = G + NMAX, LB = FC + 192, UB = LB + 16 * 20, X0 = UB + 16 * 20,
                       ACD = X0 + 16, BCD = ACD + 176, TOTAL = BCD + 160;
};

enum : int { S_OK = 0, S_MAXITER = 1, S_INFEASIBLE = 2, S_TOO_LARGE = 3, S_KKT = 4, S_WORKSET = 5 };

template <int NMAX, int HMAX>
struct Smem {
  static constexpr int QMAX = (NMAX >= 120) ? 80 : NMAX;  // working-set capacity (packed Schur inverse)
  static constexpr int NLS = NMAX / 6;
  static constexpr int MMAX = NLS * 8;
  static constexpr int RECW = ((54 + 12 * HMAX) * 4 + 2 * HMAX + 15) / 16 * 4;  // record words

  double g[NMAX];
  double Cn[2][8][6];
  double ub7[NLS];
  unsigned char vstep[NMAX], vcomp[NMAX];
  unsigned char vls[NMAX], vk[NMAX];  // variable -> its leg-step, position in it (0-2 force, 3-5 moment)
  unsigned char rmap[12 * HMAX];  // original variable -> reduced index (255 = eliminated)
  unsigned char ls_step[NLS], ls_leg[NLS], ls_vF[NLS], ls_vM[NLS];
  int n, m, nls, pad0;

  struct Asm {
    uint32_t rec[RECW];
    float sc[10][2];
    float sc234[2][2];
    float rpy[3];
    float ypsc[4];  // cy, sy, cp, sp
    float R[9], Rt[9];
    float Acd[169], Bcd[156], x0[13], W[13], Fc[192];
    float Apow[2 * 169];
    float Phi[HMAX * 156], SPhi[HMAX * 156];
    float e[13 * HMAX];
    float Hs[(NMAX / 2) * (NMAX + 1)];  // H, upper triangle, binary32 (exact), rows i and NMAX-1-i folded into one
  };
  struct Sol {
    alignas(16) double x[NMAX], xu[NMAX], z[NMAX], w[NMAX];
    alignas(16) double RS[120][NMAX / 15];   // row partials of the in-place mat-vec, one entry per lane pair
    alignas(16) double CS[240][NMAX / 30];   // mirrored (column) partials, one entry per block
    alignas(16) double piv[2][NMAX];
    double u[NMAX], d[NMAX], r[NMAX];
    double redv[NW];
    double gamma;
    int redi[NW];
    struct Rec {
      double val, raw, cn[6];
      int idx, side, vF, vM;
    } rec[NW];
    alignas(8) signed char act[MMAX];
    alignas(8) unsigned char slot[MMAX];
    unsigned char Wrow[NMAX];
    double Ep[QMAX * (QMAX + 1) / 2];  // E = (N_W M N_W')^-1, packed lower triangle: E(i,j), i>=j, at i(i+1)/2 + j
  };
  union {
    Asm a;
    Sol s;
  } u;
};

// index of H(i,j), i <= j, in the folded upper-triangle staging array: row i (< NMAX/2) and row NMAX-1-i share one
// storage row of NMAX+1 entries
template <int NMAX>
__device__ __forceinline__ int hs_index(int i, int j) {
  const bool first = 2 * i < NMAX;
  return (first ? i : NMAX - 1 - i) * (NMAX + 1) + (first ? j - i : j + 1);
}
template <int NMAX, int HMAX>
__device__ __forceinline__ double &Eref(Smem<NMAX, HMAX> &S, int i, int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  return S.u.s.Ep[hi * (hi + 1) / 2 + lo];
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
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

// ---------------------------------------------------------------------------------------------------------------
template <int NMAX, int HMAX, bool ASM_ONLY>
__global__ __launch_bounds__(NT, (NMAX >= 120 ? 2 : 3)) void hmpc_kernel(KernelArgs args) {
  using SM = Smem<NMAX, HMAX>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SM &S = *reinterpret_cast<SM *>(smem_raw);
  auto &A = S.u.a;
  auto &Q = S.u.s;

  const int tid = threadIdx.x, wv = uni(tid >> 6), ln = tid & 63;
  const int inst = ASM_ONLY ? args.dbg_index : (int)blockIdx.x;
  const int h = args.horizon;
  if (inst >= args.batch) return;
  PROF_DECL;

  // ---------------- A0: one coalesced burst brings the instance's record into LDS ----------------
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(args.records + (size_t)inst * args.stride);
    const int nwords = args.stride >> 2;
    for (int t = tid; t < nwords; t += NT) A.rec[t] = src[t];
  }
  __syncthreads();
  const float *rf = reinterpret_cast<const float *>(A.rec);
  const unsigned char *gait = reinterpret_cast<const unsigned char *>(A.rec + 54 + 12 * h);
  // record field offsets (hector_simulation_amd/records.py)
  const float *in_p = rf + 0, *in_v = rf + 3, *in_q = rf + 6, *in_w = rf + 10, *in_r = rf + 13, *in_ja = rf + 19,
              *in_wt = rf + 30, *in_al = rf + 42, *in_traj = rf + 54;

  // ---------------- A1: trigonometry, one lane per angle (SolverMPC.cpp:374-393, 333-342, 74-85) ----------------
  {
    const double PI = 3.14159265359, PI2 = 2 * PI;
    auto joint = [&](int i) -> float {
      float a = in_ja[i];
      const int k = i % 5;
      if (k == 2 || k == 4) a = (float)((double)a + 0.3 * PI);
      if (k == 3) a = (float)((double)a - 0.6 * PI);
      return (float)fmod((double)a, PI2);
    };
    if (tid < 10) {
      double s, c;
      det_sincos((double)joint(tid), s, c);
      A.sc[tid][0] = (float)s;
      A.sc[tid][1] = (float)c;
    } else if (tid < 12) {
      const int b = 5 * (tid - 10);
      float q234 = (joint(b + 2) + joint(b + 3)) + joint(b + 4);
      double s, c;
      det_sincos((double)q234, s, c);
      A.sc234[tid - 10][0] = (float)s;
      A.sc234[tid - 10][1] = (float)c;
    } else if (tid < 15) {
      const float qw = in_q[0], qx = in_q[1], qy = in_q[2], qz = in_q[3];
      if (tid == 12) {
        float n0 = 2.0f * (qw * qx + qy * qz);
        double d0 = 1.0 - (double)(2.0f * (qx * qx + qy * qy));
        A.rpy[0] = (float)det_atan2((double)n0, d0);
      } else if (tid == 13) {
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
      // swing-leg elimination tables (SolverMPC.cpp:589-637): a leg-step survives iff its Fz bound f_max*gait is not ~0
      int nv = 0, nl = 0;
      for (int i = 0; i < h; ++i) {
        float ubL = args.f_max * (float)gait[2 * i], ubR = args.f_max * (float)gait[2 * i + 1];
        const bool sL = !(ubL < 0.0001 && ubL > -.0001), sR = !(ubR < 0.0001 && ubR > -.0001);
        const int nst = (int)sL + (int)sR;
        for (int c = 0; c < 12; ++c) S.rmap[12 * i + c] = 255;
        if (nv + 6 * nst <= NMAX) {
          int k = 0;
          for (int c = 0; c < 12; ++c) {
            const int leg = (c / 3) & 1;
            if (leg == 0 ? sL : sR) {
              S.vstep[nv + k] = (unsigned char)i;
              S.vcomp[nv + k] = (unsigned char)c;
              S.rmap[12 * i + c] = (unsigned char)(nv + k);
              ++k;
            }
          }
          int rank = 0;
          for (int leg = 0; leg < 2; ++leg)
            if (leg == 0 ? sL : sR) {
              S.ls_step[nl] = (unsigned char)i;
              S.ls_leg[nl] = (unsigned char)leg;
              S.ls_vF[nl] = (unsigned char)(nv + 3 * rank);
              S.ls_vM[nl] = (unsigned char)(nv + 3 * nst + 3 * rank);
              S.ub7[nl] = (double)(leg == 0 ? ubL : ubR);
              for (int k = 0; k < 3; ++k) {
                S.vls[nv + 3 * rank + k] = (unsigned char)nl, S.vk[nv + 3 * rank + k] = (unsigned char)k;
                S.vls[nv + 3 * nst + 3 * rank + k] = (unsigned char)nl, S.vk[nv + 3 * nst + 3 * rank + k] = (unsigned char)(3 + k);
              }
              ++nl;
              ++rank;
            }
        }
        nv += 6 * nst;
      }
      S.n = nv;
      S.nls = nl;
      S.m = 8 * nl;
    }
  }
  __syncthreads();

  // ---------------- A2: scalar algebra on one lane (RobotState.cpp:17-47, SolverMPC.cpp:65-89,302-331,420-433,488-548)
  if (tid == 0) {
    const float qw = in_q[0], qx = in_q[1], qy = in_q[2], qz = in_q[3];
    float R[9], Rt[9];
    {
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
    }
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
    A.x0[12] = 9.81f;
    const float Ib[3] = {0.5413f, 0.5200f, 0.0691f};
    float RI[9], Iw[9], Iinv[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        RI[i * 3 + k] = R[i * 3 + k] * Ib[k];
        Rt[k * 3 + i] = R[i * 3 + k];
      }
    chain_mm<3, 3, 3>(RI, Rt, Iw);
    inverse3(Iw, Iinv);
    for (int i = 0; i < 9; ++i) A.R[i] = R[i], A.Rt[i] = Rt[i];

    // continuous model -> forward Euler (SolverMPC.cpp:312-331, 145-146); mass 9.0 (:423)
    const float dt = args.dt;
    for (int i = 0; i < 169; ++i) A.Acd[i] = (i % 14 == 0) ? 1.0f : 0.0f;  // fl(delta + dt*0) = delta
    for (int i = 0; i < 156; ++i) A.Bcd[i] = 0.0f;                          // fl(dt*0) = 0
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A.Acd[i * 13 + 6 + j] = 0.0f + dt * Rbi[i * 3 + j];
    for (int i = 0; i < 3; ++i) A.Acd[(3 + i) * 13 + 9 + i] = 0.0f + dt * 1.0f;
    A.Acd[11 * 13 + 12] = 0.0f + dt * -1.0f;
    const float inv_m = 1.0f / 9.0f;
    for (int leg = 0; leg < 2; ++leg) {
      const float r0 = in_r[0 + leg], r1 = in_r[2 + leg], r2 = in_r[4 + leg];
      float cm[9] = {0.0f, -r2, r1, r2, 0.0f, -r0, -r1, r0, 0.0f};
      float blk[9];
      chain_mm<3, 3, 3>(Iinv, cm, blk);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A.Bcd[(6 + i) * 12 + 3 * leg + j] = dt * blk[i * 3 + j];
          A.Bcd[(6 + i) * 12 + 6 + 3 * leg + j] = dt * Iinv[i * 3 + j];
        }
      for (int i = 0; i < 3; ++i) A.Bcd[(9 + i) * 12 + 3 * leg + i] = dt * inv_m;
    }
    for (int s = 0; s < 12; ++s) A.W[s] = in_wt[s];
    A.W[12] = 0.0f;

    // foot rotations Rz(q0)Rx(q1)Ry(q2)Ry(q3)Ry(q4) and the 16x12 constraint block (SolverMPC.cpp:426-433, 488-548)
    const float mu = 2.0f, lt = 0.09f, lh = 0.06f;
    for (int i = 0; i < 192; ++i) A.Fc[i] = 0.0f;
    for (int leg = 0; leg < 2; ++leg) {
      const int b = 5 * leg;
      const float s0 = A.sc[b][0], c0 = A.sc[b][1], s1 = A.sc[b + 1][0], c1 = A.sc[b + 1][1];
      const float s2 = A.sc[b + 2][0], c2 = A.sc[b + 2][1], s3 = A.sc[b + 3][0], c3 = A.sc[b + 3][1];
      const float s4 = A.sc[b + 4][0], c4 = A.sc[b + 4][1];
      const float s234 = A.sc234[leg][0], c234 = A.sc234[leg][1];
      float a = c0 * s2 + (c2 * s0) * s1;
      float bb = c0 * c2 - (s0 * s1) * s2;
      float d = c2 * s0 + (c0 * s1) * s2;
      float e = s0 * s2 - (c0 * c2) * s1;
      float ca3 = c3 * a + s3 * bb, sa3 = s3 * a - c3 * bb;
      float cd3 = c3 * d - s3 * e, sd3 = s3 * d + c3 * e;
      float Rf[9];
      Rf[0] = -(s4 * ca3) - c4 * sa3;
      Rf[1] = -(c1 * s0);
      Rf[2] = c4 * ca3 - s4 * sa3;
      Rf[3] = c4 * cd3 - s4 * sd3;
      Rf[4] = c0 * c1;
      Rf[5] = c4 * sd3 + s4 * cd3;
      Rf[6] = -(s234 * c1);
      Rf[7] = s1;
      Rf[8] = c234 * c1;
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
      float *row = A.Fc + (8 * leg) * 12;
      const int cf = 3 * leg, cmo = 6 + 3 * leg;
      row[0 * 12 + cf + 0] = -mu, row[0 * 12 + cf + 2] = 1.0f;
      row[1 * 12 + cf + 0] = mu, row[1 * 12 + cf + 2] = 1.0f;
      row[2 * 12 + cf + 1] = -mu, row[2 * 12 + cf + 2] = 1.0f;
      row[3 * 12 + cf + 1] = mu, row[3 * 12 + cf + 2] = 1.0f;
      for (int j = 0; j < 3; ++j) {
        row[4 * 12 + cmo + j] = t0[j];
        row[5 * 12 + cf + j] = flt[j];
        row[5 * 12 + cmo + j] = t1[j];
        row[6 * 12 + cf + j] = flh[j];
        row[6 * 12 + cmo + j] = (leg == 0) ? -t1[j] : t1[j];
      }
      row[7 * 12 + cf + 2] = 2.0f;
      // per-leg 8x6 constraint normals in binary64 (columns: F then M of this leg)
      for (int rr = 0; rr < 8; ++rr)
        for (int k = 0; k < 3; ++k) {
          S.Cn[leg][rr][k] = (double)row[rr * 12 + cf + k];
          S.Cn[leg][rr][3 + k] = (double)row[rr * 12 + cmo + k];
        }
    }
  }
  // identity power
  for (int t = tid; t < 169; t += NT) A.Apow[t] = (t % 14 == 0) ? 1.0f : 0.0f;
  __syncthreads();

  const int n = uni(S.n), m = uni(S.m), nls = uni(S.nls);
  if (n > NMAX) {  // uniform
    if (!ASM_ONLY) {
      for (int t = tid; t < 12 * h; t += NT) args.forces[(size_t)inst * 12 * h + t] = 0.0f;
      if (tid == 0) args.status[inst] = S_TOO_LARGE;
    } else if (tid == 0) {
      args.dbg_i[0] = n;
      args.dbg_i[1] = m;
    }
    return;
  }

  // ---------------- A3/A4: Acd^k by repeated right-multiplication from the identity (SolverMPC.cpp:148-158), and from each
  // power as it appears: Phi_k = Acd^k Bcd (:161-178), SPhi = fl(w_s Phi) (B'S first, as B'*S*B evaluates left to right),
  // tracking error e_i = Acd^(i+1) x0 - X_d (:457-461, :570).  Only two powers are kept (ping-pong).
  for (int k = 0; k <= h; ++k) {
    const float *Pk = A.Apow + (k & 1) * 169;
    float *Pn = A.Apow + ((k + 1) & 1) * 169;
    for (int t = tid; t < 169 + 156 + 13; t += NT) {
      if (t < 169) {
        if (k < h) {
          const int i = t / 13, j = t % 13;
          float acc = 0.0f;
#pragma unroll
          for (int mm = 0; mm < 13; ++mm) acc = ffma(Pk[i * 13 + mm], A.Acd[mm * 13 + j], acc);
          Pn[t] = acc;
        }
      } else if (t < 325) {
        if (k < h) {
          const int rem = t - 169, i = rem / 12, j = rem % 12;
          float acc = 0.0f;
#pragma unroll
          for (int mm = 0; mm < 13; ++mm) acc = ffma(Pk[i * 13 + mm], A.Bcd[mm * 12 + j], acc);
          A.Phi[k * 156 + rem] = acc;
          A.SPhi[k * 156 + rem] = A.W[i] * acc;
        }
      } else if (k >= 1) {
        const int s = t - 325, i = k - 1;
        float acc = 0.0f;
#pragma unroll
        for (int mm = 0; mm < 13; ++mm) acc = ffma(Pk[s * 13 + mm], A.x0[mm], acc);
        const float xd = (s < 12) ? in_traj[12 * i + s] : 0.0f;
        A.e[13 * i + s] = acc - xd;
      }
    }
    __syncthreads();
  }

  PROF_MARK(P_ASM);
  // ---------------- A5: g = 2 (B'S) e and H = 2(B'S B + alpha) (SolverMPC.cpp:569-570) ----------------
  if (tid < n) {
    const int a = S.vstep[tid], c = S.vcomp[tid];
    float acc = 0.0f;
    for (int i = a; i < h; ++i) {
      const float *sp = A.SPhi + (i - a) * 156 + c;
      const float *ep = A.e + 13 * i;
#pragma unroll
      for (int s = 0; s < 13; ++s) acc = ffma(sp[s * 12], ep[s], acc);
    }
    S.g[tid] = (double)(2.0f * acc);
  }
  {
    // matrix cores: 16x16 output tiles over the reduced variables, K runs over (step i ascending, state row s ascending).
    // Rows of B_qp above the block diagonal are exact zeros, which are bitwise neutral in an fmaf chain started at +0,
    // so operands before a variable's own step are fed as 0 and the chain starts at the tile's first live step.
    // The weight of state row 12 is 0 (SolverMPC.cpp:453) -> that row is neutral too and K = 12 = 3 x k4.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int nti = (n + 15) >> 4;
    const int ntiles = nti * (nti + 1) / 2;
    const int l15 = ln & 15, kq = ln >> 4;
    for (int idx = wv; idx < ntiles; idx += NW) {
      int J = 0;
      while ((J + 1) * (J + 2) / 2 <= idx) ++J;
      const int I = idx - J * (J + 1) / 2;
      const int ra = 16 * I + l15, cb = 16 * J + l15;
      const bool rav = ra < n, cbv = cb < n;
      const int sa = rav ? S.vstep[ra] : 0, ca = rav ? S.vcomp[ra] : 0;
      const int sb = cbv ? S.vstep[cb] : 0, cc = cbv ? S.vcomp[cb] : 0;
      const int istart = S.vstep[16 * J];
      f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int i = istart; i < h; ++i) {
        const bool la = rav && i >= sa, lb = cbv && i >= sb;
        const float *pa = A.SPhi + (la ? (i - sa) * 156 + ca : 0);
        const float *pb = A.Phi + (lb ? (i - sb) * 156 + cc : 0);
#pragma unroll
        for (int k4 = 0; k4 < 3; ++k4) {
          const int s = 4 * k4 + kq;
          const float av = la ? pa[s * 12] : 0.0f;
          const float bv = lb ? pb[s * 12] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int Rr = 16 * I + kq * 4 + rg, Cc = 16 * J + l15;
        if (Rr <= Cc && Cc < n) {
          const float al = (Rr == Cc) ? in_al[S.vcomp[Rr]] : 0.0f;
          const float hv = 2.0f * (acc[rg] + al);
          A.Hs[hs_index<NMAX>(Rr, Cc)] = hv;
        }
      }
    }
  }
  __syncthreads();

  PROF_MARK(P_HG);
  if (ASM_ONLY) {
    using DL = DbgLayout<NMAX>;
    float *o = args.dbg_f;
    if (tid == 0) {
      args.dbg_i[0] = n;
      args.dbg_i[1] = m;
    }
    for (int t = tid; t < n; t += NT) {
      args.dbg_i[2 + t] = 12 * S.vstep[t] + S.vcomp[t];
      o[DL::G + t] = (float)S.g[t];
    }
    for (int t = tid; t < n * n; t += NT) {
      const int i = t / n, j = t % n;
      o[DL::H + t] = A.Hs[hs_index<NMAX>(i < j ? i : j, i < j ? j : i)];
    }
    for (int t = tid; t < 192; t += NT) o[DL::FC + t] = A.Fc[t];
    const float big = 5e10f;
    for (int t = tid; t < 16 * h; t += NT) {
      const int i = t / 16, rr = t % 8, leg = (t % 16) / 8;
      float lbv, ubv;
      if (rr < 4) lbv = 0.0f, ubv = big;
      else if (rr == 4) lbv = 0.0f, ubv = 0.01f;
      else if (rr < 7) lbv = -big, ubv = 0.0f;
      else lbv = 0.0f, ubv = args.f_max * (float)gait[2 * i + leg];
      o[DL::LB + t] = lbv;
      o[DL::UB + t] = ubv;
    }
    for (int t = tid; t < 13; t += NT) o[DL::X0 + t] = A.x0[t];
    for (int t = tid; t < 169; t += NT) o[DL::ACD + t] = A.Acd[t];
    for (int t = tid; t < 156; t += NT) o[DL::BCD + t] = A.Bcd[t];
    return;
  }

  // =============================== S: M = H^-1 by symmetric sweeps, matrix in registers ===============================
  // Thread t owns the TR x TC block (rows i0.., cols j0..) of the full symmetric matrix; the 240 blocks that intersect
  // the upper triangle are enumerated block-row-major (block row tr holds block columns 2tr..29), so the lanes that
  // own pieces of one matrix row are contiguous and lanes (2k, 2k+1) always share a block row.
  // Sweep k:  d = a_kk, p = row k;  a_ij -= (p_i/d) p_j  (i,j != k);  a_kj = p_j/d;  a_kk = -1/d.  After n sweeps a = -H^-1.
  // Row k / column k entries take the same fused update with a substituted multiplier (1 - 1/d for row k, d - 1 for
  // column k: p_j - (1-1/d) p_j = p_j/d), so the inner 8x4 update has no special cases; only a_kk is patched.
  // The pivot row for sweep k+1 is published to LDS right after sweep k (double buffered) -> one barrier per sweep.
  // The inverse never leaves the registers: the whole active-set phase multiplies by it in place (rmatvec below), which
  // keeps the workgroup's LDS footprint at ~53 KB -> three workgroups (12 waves) per CU hide each other's latencies.
  constexpr int TR = NMAX / 15, TC = NMAX / 30;
  static_assert(TR * 15 == NMAX && TC * 30 == NMAX && TR == 2 * TC && TC % 2 == 0, "NMAX must be a multiple of 60");
  const bool is_v = tid < n;
  int tr = 0;
  while (tr < 14 && (tr + 1) * (30 - tr) <= tid) ++tr;
  const bool owner = tid < 240;
  const int tc = owner ? 2 * tr + (tid - tr * (31 - tr)) : 0;
  if (!owner) tr = 0;
  const int i0 = tr * TR, j0 = tc * TC;
  const int dgo = tc - 2 * tr;  // 0 or 1: the block straddles the diagonal at row offset dgo*TC; >= 2: strictly above it
  double a[TR][TC];
#pragma unroll
  for (int ii = 0; ii < TR; ++ii)
#pragma unroll
    for (int jj = 0; jj < TC; ++jj) {
      const int i = i0 + ii, j = j0 + jj;
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      a[ii][jj] = (hi < n) ? (double)A.Hs[hs_index<NMAX>(lo, hi)] : 0.0;
    }
  __syncthreads();  // every block is loaded before the solver state (which aliases the staging area) is written
  if (tid < NMAX) Q.piv[0][tid] = 0.0, Q.piv[1][tid] = 0.0;
  __syncthreads();
  if (owner && i0 == 0) {
#pragma unroll
    for (int jj = 0; jj < TC; ++jj)
      if (j0 + jj < n) Q.piv[0][j0 + jj] = a[0][jj];
  }
  __syncthreads();
  {
    const int nkb = (n + TR - 1) / TR;
    for (int kb = 0; kb < nkb; ++kb) {
      const bool rowb = owner && (tr == kb);      // my block holds matrix rows kb*TR .. kb*TR+TR-1
      const bool rown = owner && (tr == kb + 1);  // ... or the next block row (publishes its row 0 at the seam)
#pragma unroll
      for (int kk = 0; kk < TR; ++kk) {
        const int k = kb * TR + kk;
        if (k < n) {  // uniform
          const double *pv = Q.piv[k & 1];
          double *pn = Q.piv[(k + 1) & 1];
          const double d = pv[k];
          double pi[TR], pj[TC];
#pragma unroll
          for (int ii = 0; ii < TR; ii += 2) {
            const double2 t2 = *reinterpret_cast<const double2 *>(pv + i0 + ii);
            pi[ii] = t2.x, pi[ii + 1] = t2.y;
          }
#pragma unroll
          for (int jj = 0; jj < TC; jj += 2) {
            const double2 t2 = *reinterpret_cast<const double2 *>(pv + j0 + jj);
            pj[jj] = t2.x, pj[jj + 1] = t2.y;
          }
          double invd = __builtin_amdgcn_rcp(d);  // v_rcp_f64 + two Newton steps (the solver half is not bit-pinned)
          invd = dfma(dfma(-d, invd, 1.0), invd, invd);
          invd = dfma(dfma(-d, invd, 1.0), invd, invd);
          const int kc = kk % TC;                                // static column index inside the block column of k
          const bool colb = owner && (tc == 2 * kb + kk / TC);  // my block holds matrix column k
          double qi[TR];
#pragma unroll
          for (int ii = 0; ii < TR; ++ii) qi[ii] = pi[ii] * invd;
          qi[kk] = rowb ? (1.0 - invd) : qi[kk];
          pj[kc] = colb ? (d - 1.0) : pj[kc];
#pragma unroll
          for (int ii = 0; ii < TR; ++ii)
#pragma unroll
            for (int jj = 0; jj < TC; ++jj) a[ii][jj] = dfma(-qi[ii], pj[jj], a[ii][jj]);
          a[kk][kc] = (rowb && colb) ? -invd : a[kk][kc];
          // publish row k+1 of the symmetric matrix: (k+1, j>=k+1) from the row owners, (i<k+1, k+1) from the column owners
          const int k1 = k + 1;
          if (k1 < n) {
            if (kk + 1 < TR) {
              if (rowb) {
#pragma unroll
                for (int jj = 0; jj < TC; ++jj)
                  if (j0 + jj >= k1) pn[j0 + jj] = a[(kk + 1) % TR][jj];
              }
            } else {
              if (rown) {
#pragma unroll
                for (int jj = 0; jj < TC; ++jj)
                  if (j0 + jj >= k1) pn[j0 + jj] = a[0][jj];
              }
            }
            const int kc1 = (kk + 1) % TC;
            if (owner && tc == (kb * TR + kk + 1) / TC) {
#pragma unroll
              for (int ii = 0; ii < TR; ++ii)
                if (i0 + ii < k1) pn[i0 + ii] = a[ii][kc1];
            }
          }
          __syncthreads();
        }
      }
    }
  }
  // M = -a.  Lower-triangle duplicates inside diagonal blocks are zeroed and the diagonal is kept aside, so the in-place
  // mat-vec needs no masks: rows use the block as is, the mirrored (column) part subtracts the diagonal term once.
  double dg[TC];
#pragma unroll
  for (int jj = 0; jj < TC; ++jj) dg[jj] = 0.0;
#pragma unroll
  for (int ii = 0; ii < TR; ++ii)
#pragma unroll
    for (int jj = 0; jj < TC; ++jj) {
      const int df = ii - jj - dgo * TC;  // i - j for this element (only meaningful when dgo < 2)
      const bool low = owner ? (dgo < 2 && df > 0) : true;
      a[ii][jj] = low ? 0.0 : -a[ii][jj];
      if (dgo < 2 && df == 0) dg[jj] = a[ii][jj];
    }
  PROF_MARK(P_SWEEP);

  // ---- z = M w out of the register blocks.  Each thread forms the 8 row partials and 4 mirrored column partials of its
  // block; lane pairs pre-add their row partials (DPP); partials are staged in LDS and row i sums its <= 15 + 15 pieces in
  // a fixed order (deterministic).  Two barriers.  wi/wj are the vector entries at the block's rows / columns.
  auto rmv_block = [&](const double (&wi)[TR], const double (&wj)[TC]) {
    double ra[TR], ca[TC];
#pragma unroll
    for (int ii = 0; ii < TR; ++ii) {
      double s0 = 0.0;
#pragma unroll
      for (int jj = 0; jj < TC; ++jj) s0 = dfma(a[ii][jj], wj[jj], s0);
      ra[ii] = s0;
    }
#pragma unroll
    for (int jj = 0; jj < TC; ++jj) {
      double s0 = -dg[jj] * wj[jj], s1 = 0.0;
#pragma unroll
      for (int ii = 0; ii < TR; ii += 2) {
        s0 = dfma(a[ii][jj], wi[ii], s0);
        s1 = dfma(a[ii + 1][jj], wi[ii + 1], s1);
      }
      ca[jj] = s0 + s1;
    }
#pragma unroll
    for (int ii = 0; ii < TR; ++ii) ra[ii] += dpp_xor1(ra[ii]);
    if (owner) {
      if ((tid & 1) == 0) {
#pragma unroll
        for (int ii = 0; ii < TR; ii += 2) *reinterpret_cast<double2 *>(&Q.RS[tid >> 1][ii]) = make_double2(ra[ii], ra[ii + 1]);
      }
#pragma unroll
      for (int jj = 0; jj < TC; jj += 2) *reinterpret_cast<double2 *>(&Q.CS[tid][jj]) = make_double2(ca[jj], ca[jj + 1]);
    }
    __syncthreads();
    if (is_v) {
      // all loads are issued up front (clamped addresses + selects, no data-dependent trip counts): one LDS round trip
      const int btr = tid / TR, bi = tid % TR, btc = tid / TC, bj = tid % TC;
      const int p0 = (btr * (31 - btr)) >> 1, np_ = 15 - btr, nt = (btc >> 1) + 1;
      double rv[15], cv[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const int kc = (k < np_) ? k : 0;
        rv[k] = Q.RS[p0 + kc][bi];
        const int kt = (k < nt) ? k : 0;
        cv[k] = Q.CS[kt * (31 - kt) + (btc - 2 * kt)][bj];
      }
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        s0 += (k < np_) ? rv[k] : 0.0;
        s1 += (k < nt) ? cv[k] : 0.0;
      }
      Q.z[tid] = s0 + s1;
    }
    __syncthreads();
  };
  auto rmatvec_dense = [&](const double *w) {
    double wi[TR], wj[TC];
#pragma unroll
    for (int ii = 0; ii < TR; ii += 2) {
      const double2 t2 = *reinterpret_cast<const double2 *>(w + i0 + ii);
      wi[ii] = t2.x, wi[ii + 1] = t2.y;
    }
#pragma unroll
    for (int jj = 0; jj < TC; jj += 2) {
      const double2 t2 = *reinterpret_cast<const double2 *>(w + j0 + jj);
      wj[jj] = t2.x, wj[jj + 1] = t2.y;
    }
    rmv_block(wi, wj);
  };
  // vector with six non-zeros npv[0..2] at vF.., npv[3..5] at vM..: built in registers, no LDS round trip
  auto rmatvec_sparse6 = [&](const double (&npv)[6], int vF, int vM) {
    double wi[TR], wj[TC];
#pragma unroll
    for (int ii = 0; ii < TR; ++ii) {
      const int dF = i0 + ii - vF, dM = i0 + ii - vM;
      double v = 0.0;
      v = (dF == 0) ? npv[0] : v, v = (dF == 1) ? npv[1] : v, v = (dF == 2) ? npv[2] : v;
      v = (dM == 0) ? npv[3] : v, v = (dM == 1) ? npv[4] : v, v = (dM == 2) ? npv[5] : v;
      wi[ii] = v;
    }
#pragma unroll
    for (int jj = 0; jj < TC; ++jj) {
      const int dF = j0 + jj - vF, dM = j0 + jj - vM;
      double v = 0.0;
      v = (dF == 0) ? npv[0] : v, v = (dF == 1) ? npv[1] : v, v = (dF == 2) ? npv[2] : v;
      v = (dM == 0) ? npv[3] : v, v = (dM == 1) ? npv[4] : v, v = (dM == 2) ? npv[5] : v;
      wj[jj] = v;
    }
    rmv_block(wi, wj);
  };

  // =============================== Q: dual active set (Goldfarb-Idnani, range-space form) ===============================
  // Fixed thread roles: thread c < m = constraint row c (leg-step c>>3, row c&7) with its 6 coefficients, variable
  // offsets and bounds in registers; thread i < n = variable i.
  const double INF = __builtin_huge_val();
  const double FEAS_TOL = 1e-9;
  const bool is_c = tid < m;
  int c_vF = 0, c_vM = 0;
  double c_cn[6] = {0, 0, 0, 0, 0, 0}, c_ub = INF, c_scale = 1.0;
  bool c_hasl = false, c_hasu = false;
  if (is_c) {
    const int e = tid >> 3, rr = tid & 7, leg = S.ls_leg[e];
    c_vF = S.ls_vF[e], c_vM = S.ls_vM[e];
#pragma unroll
    for (int k = 0; k < 6; ++k) c_cn[k] = S.Cn[leg][rr][k];
    c_hasl = (rr <= 4) || (rr == 7);  // every finite lower bound is 0 (SolverMPC.cpp:466-482)
    c_hasu = (rr >= 4);
    c_ub = (rr == 4) ? (double)0.01f : (rr == 7 ? S.ub7[e] : 0.0);
    c_scale = (rr == 7 && c_ub > 1.0) ? 1.0 / c_ub : 1.0;  // the Fz cap is O(f_max): compare it on a unit scale
  }
  int v_e = 0, v_k = 0, v_leg = 0;
  if (is_v) {
    v_e = S.vls[tid], v_k = S.vk[tid];
    v_leg = S.ls_leg[v_e];
  }
  for (int t = tid; t < m; t += NT) {
    Q.act[t] = 0;
    Q.slot[t] = 0;
  }
  if (tid < NMAX) Q.r[tid] = 0.0;
  if (tid < NMAX) Q.w[tid] = is_v ? -S.g[tid] : 0.0;  // entries >= n stay exactly 0
  __syncthreads();
  rmatvec_dense(Q.w);  // unconstrained minimiser x_u = -M g
  if (is_v) {
    const double xv = Q.z[tid];
    Q.xu[tid] = xv;
    Q.x[tid] = xv;
  }
  __syncthreads();
  PROF_MARK(P_XU);

  int q = 0, iters = 0, code = S_OK;
  const int itmax = 4 * m + 16;

  // slack of this thread's constraint row on its tighter side at xv (unit-scaled); side = +1 lower, -1 upper
  auto my_slack = [&](const double *xv, int &side, double &raw) -> double {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) s0 = dfma(c_cn[k], xv[c_vF + k], s0);
#pragma unroll
    for (int k = 0; k < 3; ++k) s1 = dfma(c_cn[3 + k], xv[c_vM + k], s1);
    const double s = s0 + s1;
    const double sl = c_hasl ? s : INF;
    const double su = c_hasu ? (c_ub - s) : INF;
    const double ssu = su * c_scale;
    side = (sl <= ssu) ? 1 : -1;
    raw = (sl <= ssu) ? sl : su;
    return (sl <= ssu) ? sl : ssu;
  };
  // w_i = sum over the active rows of variable i's leg-step of coef(row) * a_row[i]  (+ extra)
  auto gather_w = [&](const double *coefv, double sgn, double extra) {
    if (is_v) {
      const unsigned long long am = *reinterpret_cast<const unsigned long long *>(&Q.act[8 * v_e]);
      const unsigned long long sm = *reinterpret_cast<const unsigned long long *>(&Q.slot[8 * v_e]);
      double acc0 = extra, acc1 = 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int ac = (int)(signed char)((am >> (8 * rr)) & 0xff);
        const int sl = (int)((sm >> (8 * rr)) & 0xff);
        const double cf = coefv[sl];
        const double coef = (ac == 0) ? 0.0 : ((ac > 0) ? sgn * cf : -sgn * cf);
        const double cv = S.Cn[v_leg][rr][v_k];
        if (rr & 1) acc1 = dfma(coef, cv, acc1);
        else acc0 = dfma(coef, cv, acc0);
      }
      Q.w[tid] = acc0 + acc1;
    }
  };

  for (int pass = 0; pass < 3 && code == S_OK; ++pass) {
    // ---- main loop ----
    while (true) {
      // (1) most violated constraint; the winning lane of each wave also publishes its constants
      double val = INF, raw = INF;
      int side = 1;
      if (is_c && Q.act[tid] == 0) val = my_slack(Q.x, side, raw);
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
          rc.vF = c_vF;
          rc.vM = c_vM;
#pragma unroll
          for (int k = 0; k < 6; ++k) rc.cn[k] = c_cn[k];
        }
      }
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
      if (iters >= itmax) {
        code = S_MAXITER;
        break;
      }
      wsel = uni(wsel);
      const int p = uni(Q.rec[wsel].idx), sgi = uni(Q.rec[wsel].side), vFp = uni(Q.rec[wsel].vF), vMp = uni(Q.rec[wsel].vM);
      const int ep = p >> 3;
      double sp = Q.rec[wsel].raw;
      const double sg = (double)sgi;
      double np[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) np[k] = sg * Q.rec[wsel].cn[k];
      double up = 0.0;
      bool added = false;
      while (!added) {
        ++iters;
        if (iters > itmax) {
          code = S_MAXITER;
          break;
        }
        // (2) y = M n+ in place (n+ has six non-zeros), then every constraint thread forms a_c' y; active rows scatter
        //     d[slot] = sign * a_c' y; row p gives gamma = n+' M n+
        rmatvec_sparse6(np, vFp, vMp);
        if (is_c) {
          double d0 = 0.0, d1 = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) d0 = dfma(c_cn[k], Q.z[c_vF + k], d0);
#pragma unroll
          for (int k = 0; k < 3; ++k) d1 = dfma(c_cn[3 + k], Q.z[c_vM + k], d1);
          const double dc = d0 + d1;
          const int ac = Q.act[tid];
          if (ac != 0) Q.d[Q.slot[tid]] = (double)ac * dc;
          if (tid == p) Q.gamma = sg * dc;
        }
        __syncthreads();
        PROF_MARK(P_D);
        // (3) r = E d : 4 lanes per row, quad reduction; the lane that owns r_j also bids for the dual step length
        //     t1 = min_j u_j / r_j over r_j > 0 (largest step keeping u >= 0)
        {
          double t1c = INF;
          int t1j = 0;
          for (int jb = 0; jb < q; jb += NT / 4) {
            const int j = jb + (tid >> 2), part = tid & 3;
            double acc = 0.0;
            if (j < q) {
              double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
              for (int i = part; i < q; i += 16) {
                const int i1 = (i + 4 < q) ? i + 4 : i, i2 = (i + 8 < q) ? i + 8 : i, i3 = (i + 12 < q) ? i + 12 : i;
                const double e0 = Eref(S, j, i), e1 = Eref(S, j, i1), e2 = Eref(S, j, i2), e3 = Eref(S, j, i3);
                const double d0 = Q.d[i], d1 = Q.d[i1], d2 = Q.d[i2], d3 = Q.d[i3];
                a0 = dfma(e0, d0, a0);
                a1 = (i + 4 < q) ? dfma(e1, d1, a1) : a1;
                a2 = (i + 8 < q) ? dfma(e2, d2, a2) : a2;
                a3 = (i + 12 < q) ? dfma(e3, d3, a3) : a3;
              }
              acc = (a0 + a1) + (a2 + a3);
            }
            acc += dpp_xor1(acc);
            acc += dpp_xor2(acc);
            if (j < q && part == 0) {
              Q.r[j] = acc;
              if (acc > 1e-14) {
                const double tj = Q.u[j] / acc;
                if (tj < t1c) t1c = tj, t1j = j;
              }
            }
          }
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
        rmatvec_dense(Q.w);
        PROF_MARK(P_MV);
        // (6) step lengths and the step
        double delta = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) delta = dfma(np[k], Q.z[vFp + k], delta);
#pragma unroll
        for (int k = 0; k < 3; ++k) delta = dfma(np[3 + k], Q.z[vMp + k], delta);
        const double gamma = Q.gamma;
        int l = Q.redi[0];
        double t1 = Q.redv[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const double ov = Q.redv[w];
          if (ov < t1) t1 = ov, l = Q.redi[w];
        }
        l = uni(l);
        const bool dep = ub(!(delta > 1e-12 * gamma));
        const double t2 = dep ? INF : -sp / delta;
        const double t = (t1 < t2) ? t1 : t2;
        if (ub(t == INF)) {
          code = S_INFEASIBLE;
          break;
        }
        if (!dep && is_v) Q.x[tid] = dfma(t, Q.z[tid], Q.x[tid]);
        if (tid < q) Q.u[tid] = dfma(-t, Q.r[tid], Q.u[tid]);
        up += t;
        if (!dep) sp = dfma(t, delta, sp);
        const bool fullstep = ub(!dep && !(t1 < t2));
        if (fullstep && q >= SM::QMAX) {
          code = S_WORKSET;
          break;
        }
        if (fullstep) {
          // full step: constraint p joins the working set; bordered update of E (16 x 16 thread tiles of the lower triangle)
          const double idl = 1.0 / delta;
          const int ti = tid >> 4, tj = tid & 15;
          for (int ib = 0; ib < q; ib += 16) {
            const int i = ib + ti;
            if (i < q) {
              const double ri = Q.r[i] * idl;
              for (int j = tj; j <= i; j += 16) {
                double &ee = Q.Ep[i * (i + 1) / 2 + j];
                ee = dfma(ri, Q.r[j], ee);
              }
            }
          }
          if (tid < q) Q.Ep[q * (q + 1) / 2 + tid] = -Q.r[tid] * idl;
          if (tid == q) {
            Q.Ep[q * (q + 1) / 2 + q] = idl;
            Q.u[q] = up;
            Q.Wrow[q] = (unsigned char)p;
            Q.act[p] = (signed char)sgi;
            Q.slot[p] = (unsigned char)q;
          }
          ++q;
          added = true;
          __syncthreads();
        } else {
          // partial (or pure dual) step: slot l leaves; Schur-complement downdate of E (row/column l are read-only
          // during the pass, so no staging copy is needed), then the last slot moves into l
          const double iel = 1.0 / Eref(S, l, l);
          const int ti = tid >> 4, tj = tid & 15;
          for (int ib = 0; ib < q; ib += 16) {
            const int i = ib + ti;
            if (i < q && i != l) {
              const double ci = Eref(S, i, l) * iel;
              for (int j = tj; j <= i; j += 16) {
                if (j != l) {
                  double &ee = Q.Ep[i * (i + 1) / 2 + j];
                  ee = dfma(-ci, Eref(S, j, l), ee);
                }
              }
            }
          }
          __syncthreads();
          const int last = q - 1;
          if (l != last) {
            if (tid < last && tid != l) Eref(S, l, tid) = Eref(S, last, tid);
            if (tid == l) Eref(S, l, l) = Eref(S, last, last);
          }
          if (tid == 0) {
            const int cl = Q.Wrow[l];
            Q.act[cl] = 0;
            if (l != last) {
              const int cm = Q.Wrow[last];
              Q.Wrow[l] = (unsigned char)cm;
              Q.slot[cm] = (unsigned char)l;
              Q.u[l] = Q.u[last];
            }
          }
          --q;
          __syncthreads();
        }
        PROF_MARK(P_UPD);
      }
      if (code != S_OK) break;
    }
    if (code != S_OK || q == 0) break;

    // ---- refinement of the multipliers on the final working set: u += E (b_W - N_W x(u)), x(u) = x_u + M N_W' u ----
    for (int it = 0; it < 3; ++it) {
      gather_w(Q.u, 1.0, 0.0);
      __syncthreads();
      rmatvec_dense(Q.w);
      if (is_v) Q.x[tid] = Q.xu[tid] + Q.z[tid];
      __syncthreads();
      if (it == 2) break;
      if (is_c) {
        const int ac = Q.act[tid];
        if (ac != 0) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) s = dfma(c_cn[k], Q.x[c_vF + k], s);
#pragma unroll
          for (int k = 0; k < 3; ++k) s = dfma(c_cn[3 + k], Q.x[c_vM + k], s);
          const double bnd = (ac > 0) ? 0.0 : c_ub;
          Q.d[Q.slot[tid]] = (double)ac * (bnd - s);  // b_j - n_j' x with n_j = sign*a_j, b_j = sign*bound
        }
      }
      __syncthreads();
      for (int jb = 0; jb < q; jb += NT / 4) {
        const int j = jb + (tid >> 2), part = tid & 3;
        double acc = 0.0;
        if (j < q) {
          double a0 = 0.0, a1 = 0.0;
          for (int i = part; i < q; i += 8) {
            const int i1 = (i + 4 < q) ? i + 4 : i;
            const double e0 = Eref(S, j, i), e1 = Eref(S, j, i1);
            const double d0 = Q.d[i], d1 = Q.d[i1];
            a0 = dfma(e0, d0, a0);
            a1 = (i + 4 < q) ? dfma(e1, d1, a1) : a1;
          }
          acc = a0 + a1;
        }
        acc += dpp_xor1(acc);
        acc += dpp_xor2(acc);
        if (j < q && part == 0) Q.u[j] += acc;
      }
      __syncthreads();
    }
    // a refinement that moved x across another constraint sends us back into the main loop (rare)
  }

  PROF_MARK(P_POLISH);
  // final KKT check: primal slack and multiplier signs
  __syncthreads();
  {
    double val = INF, raw;
    int side;
    if (is_c && Q.act[tid] == 0) val = my_slack(Q.x, side, raw);
    double umin = (tid < q) ? Q.u[tid] : INF;
    val = wave_min(val);
    umin = wave_min(umin);
    if (ln == 0) Q.redv[wv] = val, Q.rec[wv].raw = umin;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      val = (Q.redv[w] < val) ? Q.redv[w] : val;
      umin = (Q.rec[w].raw < umin) ? Q.rec[w].raw : umin;
    }
    if (code == S_OK && (val < -1e-6 || umin < -1e-6)) code = S_KKT;
  }


  // ---------------- output: scatter to the reference's 12h layout, eliminated variables exactly 0 (SolverMPC.cpp:720-732)
  for (int t = tid; t < 12 * h; t += NT) {
    const int rmp = S.rmap[t];
    const double xv = (rmp == 255) ? 0.0 : Q.x[rmp];
    args.forces[(size_t)inst * 12 * h + t] = (float)xv;
    if (args.x64) args.x64[(size_t)inst * 12 * h + t] = xv;
  }
  if (tid == 0) {
    args.status[inst] = (uint32_t)code | ((uint32_t)(iters & 0xfff) << 8) | ((uint32_t)(q & 0xfff) << 20);
    if (args.obj64) {
      // objective through the KKT identity  0.5 x'Hx + g'x = 0.5 g'x + 0.5 u'b_W  (H itself was consumed by the sweeps)
      double o = 0.0;
      for (int i = 0; i < n; ++i) o = dfma(0.5 * S.g[i], Q.x[i], o);
      for (int j = 0; j < q; ++j) {
        const int c = Q.Wrow[j], rr = c & 7;
        const double sj = (double)Q.act[c];
        const double bnd = (sj > 0) ? 0.0 : ((rr == 4) ? (double)0.01f : (rr == 7 ? S.ub7[c >> 3] : 0.0));
        o = dfma(0.5 * Q.u[j], sj * bnd, o);
      }
      args.obj64[inst] = o;
    }
  }
  PROF_MARK(P_FINAL);
  PROF_FLUSH();
}

}  // namespace hmpc
