// hmpc_math.h -- deterministic scalar math for the assembly stage (device side, gfx950).
//
// The assembly of the MPC QP must be bit-reproducible (cond(H) ~ 2.4e6, so two legitimate
// fp32 evaluation orders move the optimal forces by up to 7.5e-4).  The arithmetic contract ("HMPC-A1", DESIGN.md
// section 3) is: IEEE binary32 round-to-nearest-even, no implicit contraction (this translation unit is compiled with
// -ffp-contract=off), every contraction a k-ascending fmaf chain started at +0, and trigonometry evaluated by the
// binary64 routines below (only +,-,*,/,sqrt,rint and explicit fma, each a correctly rounded IEEE operation) and
// rounded once to binary32.  What the routines compute for the reference: cos/sin in euler_to_rotation and the foot
// rotations (ConvexMPC/SolverMPC.cpp:74-85, 428-433), atan2/asin in quat_to_rpy (SolverMPC.cpp:338-341).
#pragma once
#include <hip/hip_runtime.h>

namespace hmpc {

__device__ __forceinline__ double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Cody-Waite split of pi/2 (33 + 33 + 53 bits) and Taylor coefficients (-1)^k/(2k+1)!, (-1)^k/(2k)!, (-1)^k/(2k+1)
#define HMPC_PIO2_1 0x1.921fb54400000p+0
#define HMPC_PIO2_2 0x1.0b4611a600000p-34
#define HMPC_PIO2_3 0x1.3198a2e037073p-69
#define HMPC_TWO_OVER_PI 0x1.45f306dc9c883p-1
#define HMPC_PIO2 0x1.921fb54442d18p+0
#define HMPC_PI 0x1.921fb54442d18p+1

__device__ inline void det_sincos(double x, double &s, double &c) {
  const double SC[9] = {-0x1.5555555555555p-3,  0x1.1111111111111p-7,  -0x1.a01a01a01a01ap-13,
                        0x1.71de3a556c734p-19,  -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33,
                        -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49,  -0x1.2f49b46814157p-57};
  const double CC[9] = {-0x1.0000000000000p-1,  0x1.5555555555555p-5,  -0x1.6c16c16c16c17p-10,
                        0x1.a01a01a01a01ap-16,  -0x1.27e4fb7789f5cp-22, 0x1.1eed8eff8d898p-29,
                        -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45,  -0x1.6827863b97d97p-53};
  double k = __builtin_rint(x * HMPC_TWO_OVER_PI);
  double r = dfma(-k, HMPC_PIO2_1, x);
  r = dfma(-k, HMPC_PIO2_2, r);
  r = dfma(-k, HMPC_PIO2_3, r);
  double z = r * r;
  double ps = SC[8], pc = CC[8];
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    ps = dfma(ps, z, SC[i]);
    pc = dfma(pc, z, CC[i]);
  }
  double sn = dfma(r * z, ps, r);
  double cs = dfma(z, pc, 1.0);
  long long q = (long long)k & 3;
  if (q == 0) {
    s = sn, c = cs;
  } else if (q == 1) {
    s = cs, c = -sn;
  } else if (q == 2) {
    s = -sn, c = -cs;
  } else {
    s = -cs, c = sn;
  }
}

__device__ inline double det_atan01(double t) {
  const double AC[8] = {-0x1.5555555555555p-2, 0x1.999999999999ap-3, -0x1.2492492492492p-3, 0x1.c71c71c71c71cp-4,
                        -0x1.745d1745d1746p-4, 0x1.3b13b13b13b14p-4, -0x1.1111111111111p-4, 0x1.e1e1e1e1e1e1ep-5};
  const double TAB[9] = {0x0.0p+0,
                         0x1.fd5ba9aac2f6ep-4,
                         0x1.f5b75f92c80ddp-3,
                         0x1.6f61941e4def1p-2,
                         0x1.dac670561bb4fp-2,
                         0x1.1e00babdefeb4p-1,
                         0x1.4978fa3269ee1p-1,
                         0x1.700a7c5784634p-1,
                         0x1.921fb54442d18p-1};
  double fi = __builtin_rint(t * 8.0);
  int i = (int)fi;
  double cpt = fi * 0.125;
  double u = (t - cpt) / dfma(t, cpt, 1.0);
  double z = u * u;
  double p = AC[7];
#pragma unroll
  for (int k = 6; k >= 0; --k) p = dfma(p, z, AC[k]);
  double a = dfma(u * z, p, u);
  double tab = TAB[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) tab = (i == k) ? TAB[k] : tab;  // select chain instead of a private-memory table
  return tab + a;
}

__device__ inline double det_atan2(double y, double x) {
  double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
  double a;
  if (ax == 0.0 && ay == 0.0) {
    a = 0.0;
  } else if (ay <= ax) {
    a = det_atan01(ay / ax);
  } else {
    a = HMPC_PIO2 - det_atan01(ax / ay);
  }
  if (x < 0.0) a = HMPC_PI - a;
  return (y < 0.0) ? -a : a;
}

__device__ inline double det_asin(double v) { return det_atan2(v, __builtin_sqrt((1.0 - v) * (1.0 + v))); }

// 3x3 inverse by the adjugate (the closed form Eigen applies to fixed 3x3, SolverMPC.cpp:87 and :320)
__device__ inline void inverse3(const float *m, float *inv) {
  float cof[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float a = m[i1 * 3 + j1] * m[i2 * 3 + j2];
      float b = m[i1 * 3 + j2] * m[i2 * 3 + j1];
      cof[i * 3 + j] = a - b;
    }
  float d0 = cof[0] * m[0], d1 = cof[3] * m[3], d2 = cof[6] * m[6];
  float det = (d0 + d1) + d2;
  float invdet = 1.0f / det;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) inv[r * 3 + c] = cof[c * 3 + r] * invdet;
}

// out(MxN) = A(MxK) B(KxN), row-major, k-ascending fmaf chain from +0
template <int M, int K, int N>
__device__ inline void chain_mm(const float *A, const float *B, float *out) {
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc = ffma(A[i * K + k], B[k * N + j], acc);
      out[i * N + j] = acc;
    }
}

}  // namespace hmpc
