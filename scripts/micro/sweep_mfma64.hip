// Developer prototype: the inverse of a 120x120 SPD matrix by 4x4 block-pivot symmetric Gauss-Jordan sweeps with the
// rank-4 updates on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), the matrix resident in the accumulator registers
// (36 upper-triangle 16x16 tiles over 4 waves = 9 tiles = 72 VGPRs per lane -- the footprint of the scalar sweeps' 6x6 block).
// Measures cycles per workgroup with three workgroups per CU (the headline variant's occupancy) and the accuracy of the
// inverse; the scalar sweeps of hmpc_kernel.h cost ~122 k cycles per workgroup in the same setting (profiles/r03).
// This is the stand-alone prototype of mfma_sweeps() in hmpc_kernel.h (which adds the power-of-two scaling and the hand-over
// to the 6 x 6 blocks).  -DDESYNC staggers the workgroups (identical workgroups started together run in lockstep and never
// overlap their matrix instructions with each other's latency chains: 101 k; staggered as in the product: 78 k);
// -DNO_MFMA leaves the matrix instructions out (what the rest costs).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/sweep_mfma64.hip -o /tmp/sweep_mfma64 && /tmp/sweep_mfma64
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#ifdef LOOKAHEAD
// -DLOOKAHEAD (round 5): the pivot block's inverse leaves the critical path.  The 36 tiles are dealt to THREE waves (12 each);
// the fourth wave holds a private copy of the diagonal tile that hosts the NEXT pivot block, applies the step's update to it
// first (one matrix instruction, same operands as the owner's: the same bits), picks the next pivot block out of it and runs the
// LDL' while the other three waves are still in their matrix instructions; the owner of a diagonal tile hands it over through
// LDS once every four steps.
constexpr int N = 120, NP = 128, NT = 256, NTI = 8, TPW = 12;  // tiles per (tile-holding) wave
#else
constexpr int N = 120, NP = 128, NT = 256, NTI = 8, TPW = 9;  // tiles per wave
#endif
constexpr int PST = NP + 8;                                   // panel row stride (doubles)

__device__ __forceinline__ double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = dfma(dfma(-d, r, 1.0), r, r);
  r = dfma(dfma(-d, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double rcp1(double d) {  // v_rcp_f64 (2^-24) + one Newton step: 2e-15
  double r = __builtin_amdgcn_rcp(d);
  return dfma(dfma(-d, r, 1.0), r, r);
}

struct Smem {
  double P[2][4][PST];  // published pivot panel rows (double buffered); the K columns carry D - I
  double Dinv[2][4][4]; // inverse of the step's 4x4 pivot block, published with the panel by the wave that owns the diagonal tile
  double Draw[4][4];    // the pivot block as it is (D recovered from the panel's D - I would cost small pivots three digits)
  double DT[256];       // LOOKAHEAD: hand-over of a diagonal tile to the pivot wave, register r of lane l at 64 r + l
  double pad[5200 - 256];     // brings the footprint to the headline variant's (three workgroups per CU by LDS as well)
};

// tile t (0..35, block-row-major over I <= J) -> (I, J)
constexpr int tile_i(int t) {
  int i = 0, base = 0;
  while (t >= base + (NTI - i)) base += NTI - i, ++i;
  return i;
}
constexpr int tile_j(int t) {
  int i = 0, base = 0;
  while (t >= base + (NTI - i)) base += NTI - i, ++i;
  return i + (t - base);
}

// one wave's share of the sweeps; WV (its index in the workgroup) is a template parameter so that the tile coordinates are
// compile-time constants: LDS addresses become immediate offsets, no address registers, no coordinate tables
__device__ __forceinline__ void ldl_dinv_row(Smem &S, const int s, const int g, const int c);

#ifdef LOOKAHEAD
// the pivot wave (wave 3)
__device__ __forceinline__ void pivot_wave(Smem &S, long long *cyc, int n) {
  const int tid = threadIdx.x, ln = tid & 63, g = ln >> 4, c = ln & 15;
#ifdef DESYNC
  for (int d = 0; d < (int)((blockIdx.x * 37u) % 61u) * 64; ++d) __builtin_amdgcn_s_sleep(1);
  __syncthreads();
#endif
  d4 dt;
  auto pick = [&](const d4 &v, int rr) __attribute__((always_inline)) -> double {
    const double lo = (rr & 1) ? v[1] : v[0], hi = (rr & 1) ? v[3] : v[2];
    return (rr & 2) ? hi : lo;
  };
  auto load_dt = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dt[r] = S.DT[64 * r + ln];
  };
  // pivot block of step sp out of the private tile -> Draw -> row g of its inverse -> Dinv[sp & 1]
  auto invert = [&](const int sp) __attribute__((always_inline)) {
    const int rr = sp & 3, c0 = 4 * rr;
    const double v = pick(dt, rr);
    if (c >= c0 && c < c0 + 4) S.Draw[g][c - c0] = v;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    ldl_dinv_row(S, sp, g, c);
  };
  __syncthreads();  // (prologue: the owner of tile (0, 0) has handed it over)
  load_dt();
  invert(0);
  __syncthreads();
  const int nsteps = (n + 3) >> 2;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) {
      const int Ik1 = (s + 1) >> 2;
      if (((s + 1) & 3) == 0) load_dt();  // a new diagonal tile, handed over at the end of step s - 1
      const double(*P)[PST] = S.P[s & 1];
      const double x0 = S.Dinv[s & 1][g][0], x1 = S.Dinv[s & 1][g][1], x2 = S.Dinv[s & 1][g][2], x3 = S.Dinv[s & 1][g][3];
      const int m = 16 * Ik1 + c;
      const double a = -dfma(x3, P[3][m], dfma(x2, P[2][m], dfma(x1, P[1][m], x0 * P[0][m])));
      const double b = P[g][m];
      dt = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, dt, 0, 0, 0);
      invert(s + 1);
    }
    __syncthreads();
  }
}
#endif

template <int WV>
__device__ __forceinline__ void sweep_wave(Smem &S, const double *H, double *M, long long *cyc, int n) {
  const int tid = threadIdx.x, ln = tid & 63, g = ln >> 4, c = ln & 15;
  // wave w owns tiles 9 w .. 9 w + 8 of the block-row-major numbering: at most four distinct tile rows per wave
  struct TI {
    int v[TPW];
    constexpr TI(bool col) : v{} {
      for (int t = 0; t < TPW; ++t) v[t] = col ? tile_j(TPW * WV + t) : tile_i(TPW * WV + t);
    }
    constexpr int operator[](int t) const { return v[t]; }
  };
  constexpr TI tI(false), tJ(true);
  d4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * tI[t] + g + 4 * r, j = 16 * tJ[t] + c;
      acc[t][r] = (i < n && j < n) ? H[i * N + j] : ((i == j) ? 1.0 : 0.0);  // identity padding
    }
  }
#ifdef DESYNC
  // identical workgroups started together run in lockstep (all in their matrix instructions at once, then all in their
  // latency chains at once) -- unlike the product kernel, whose workgroups sit at unrelated phases.  A start-up delay that
  // differs per workgroup breaks the lockstep.
  for (int d = 0; d < (int)((blockIdx.x * 37u) % 61u) * 64; ++d) __builtin_amdgcn_s_sleep(1);
  __syncthreads();
#endif
  const long long t0 = clock64();
  // panel of step s from the accumulators: rows K = 4 s .. 4 s + 3 of the symmetric matrix, the pivot block as D - I.
  //   row tiles (Ik, J):      lane (g, c) holds A[16 Ik + 4 rr + g][16 J + c] in register rr         (rr = s % 4)
  //   column tiles (I < Ik):  lanes with c in [4 rr, 4 rr + 4) hold A[16 I + g + 4 r][16 Ik + c], r = 0..3
  auto pick = [&](const d4 &v, int rr) __attribute__((always_inline)) -> double {  // rr uniform: scalar-conditioned selects
    const double lo = (rr & 1) ? v[1] : v[0], hi = (rr & 1) ? v[3] : v[2];
    return (rr & 2) ? hi : lo;
  };
  // x = row g of D^-1 (D = [A B; B' C] in 2x2 blocks; the panel carries D - I), every lane for its own g:
  //   S = C - B' A^-1 B,  x_lo = S^-1 (v - B' A^-1 u),  x_hi = A^-1 u - (A^-1 B) x_lo      for e_g = [u; v]
  // Computed ONCE per step, by the wave that owns the diagonal tile, right after it has published the panel (its own LDS
  // writes are visible to it after a wait): lanes c == 0 publish the four rows.
  auto publish_dinv = [&](const int s) __attribute__((always_inline)) { ldl_dinv_row(S, s, g, c); };
  // One computed jump on the tile row of the pivot instead of two tests per tile: inside a case the tile coordinates AND the
  // pivot's tile row are compile-time constants, so only the tiles that really hold panel entries leave code behind.
  auto publish_ik = [&](auto ikc, const int s, const int rr) __attribute__((always_inline)) {
    constexpr int IK = decltype(ikc)::value;
    const int c0 = 4 * rr;
    double(*P)[PST] = S.P[s & 1];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (tI[t] == IK) {  // compile time
        double v = pick(acc[t], rr);
        if (tJ[t] == IK) {
#ifndef LOOKAHEAD
          if (c >= c0 && c < c0 + 4) S.Draw[g][c - c0] = v;
#endif
          v -= (c == c0 + g) ? 1.0 : 0.0;
        }
        P[g][16 * tJ[t] + c] = v;
#ifndef LOOKAHEAD
        if (tJ[t] == IK) {  // this wave owns the diagonal tile = the pivot block
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes have landed
          publish_dinv(s);
        }
#endif
      } else if (tJ[t] == IK) {
        if (c >= c0 && c < c0 + 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) P[c - c0][16 * tI[t] + g + 4 * r] = acc[t][r];
        }
      }
    }
  };
  auto handover = [&](const int sp) __attribute__((always_inline)) {  // sp = -1: the prologue (tile (0, 0) as it is)
#ifdef LOOKAHEAD
    if (((sp + 1) & 3) == 0) {
      const int In = (sp + 1) >> 2;
#pragma unroll
      for (int t = 0; t < TPW; ++t)
        if (tI[t] == tJ[t]) {  // compile time
          if (tI[t] == In) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S.DT[64 * r + ln] = acc[t][r];
          }
        }
    }
#endif
  };
  auto publish = [&](const int s, const int rr) __attribute__((always_inline)) {
    handover(s);
    switch (s >> 2) {  // uniform
      case 0: publish_ik(std::integral_constant<int, 0>(), s, rr); break;
      case 1: publish_ik(std::integral_constant<int, 1>(), s, rr); break;
      case 2: publish_ik(std::integral_constant<int, 2>(), s, rr); break;
      case 3: publish_ik(std::integral_constant<int, 3>(), s, rr); break;
      case 4: publish_ik(std::integral_constant<int, 4>(), s, rr); break;
      case 5: publish_ik(std::integral_constant<int, 5>(), s, rr); break;
      case 6: publish_ik(std::integral_constant<int, 6>(), s, rr); break;
      default: publish_ik(std::integral_constant<int, 7>(), s, rr); break;
    }
  };
  auto step = [&](const int s, const int rr) __attribute__((always_inline)) {  // rr = s % 4
    const int Ik = s >> 2;
    const double(*P)[PST] = S.P[s & 1];
    const double x0 = S.Dinv[s & 1][g][0], x1 = S.Dinv[s & 1][g][1], x2 = S.Dinv[s & 1][g][2], x3 = S.Dinv[s & 1][g][3];
    // rank-4 updates: tile(I,J) -= Q_I' P_J,  Q = D^-1 P;  A operand: lane (g, c) supplies -Q[g][16 I + c], B: P[g][16 J + c].
    // Operands first (every LDS read in flight before the first use), then the matrix instructions back to back.
    // Groups of three tiles, software pipelined: the operands of group k+1 are read while the matrix instructions of group k run.
    constexpr int GT = 3, NGRP = TPW / GT;
    double aop[2][GT], bop[2][GT];
    double alast = 0.0;
    auto fetch = [&](const int grp, double (&ao)[GT], double (&bo)[GT]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int t = grp * GT + u;
        bo[u] = P[g][16 * tJ[t] + c];
        if (t == 0 || tI[t] != tI[t - 1]) {  // uniform; the wave's tiles are sorted by I
          const int m = 16 * tI[t] + c;
          alast = -dfma(x3, P[3][m], dfma(x2, P[2][m], dfma(x1, P[1][m], x0 * P[0][m])));
        }
        ao[u] = alast;
      }
    };
    fetch(0, aop[0], bop[0]);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
      if (grp + 1 < NGRP) fetch(grp + 1, aop[(grp + 1) & 1], bop[(grp + 1) & 1]);
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int t = grp * GT + u;
#ifndef NO_MFMA
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[grp & 1][u], bop[grp & 1][u], acc[t], 0, 0, 0);
#else
        acc[t][0] += aop[grp & 1][u] * bop[grp & 1][u];
#endif
      }
    }
    // the pivot block came out as 2 I - D^-1 (substituted multipliers on both sides): its diagonal is 2 too high
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      if (tI[t] == tJ[t] && tI[t] == Ik) {  // (first test compile time, second uniform)
        asm volatile("");
        const double two = (c == 4 * rr + g) ? 2.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] -= (r == rr) ? two : 0.0;
      }
  };
#ifdef LOOKAHEAD
  handover(-1);
  __syncthreads();
  {  // P(0) (publish() would hand tile 1 over at s = 3 only: handover(0) is a no-op)
    publish(0, 0);
  }
  __syncthreads();
#else
  publish(0, 0);
  __syncthreads();
#endif
  const int nsteps = (n + 3) >> 2;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    step(s, s & 3);
    if (s + 1 < nsteps) publish(s + 1, (s + 1) & 3);
    __syncthreads();
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * tI[t] + g + 4 * r, j = 16 * tJ[t] + c;
      if (i < n && j < n) {
        M[i * N + j] = -acc[t][r];
        if (tI[t] != tJ[t]) M[j * N + i] = -acc[t][r];
      }
    }
}

__global__ __launch_bounds__(NT, 3) void sweep_kernel(const double *Hin, double *Mout, long long *cyc, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  Smem &S = *reinterpret_cast<Smem *>(raw);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double *H = Hin + (size_t)blockIdx.x * N * N;
  double *M = Mout + (size_t)blockIdx.x * N * N;
  switch (wv) {  // uniform
    case 0: sweep_wave<0>(S, H, M, cyc, n); break;
    case 1: sweep_wave<1>(S, H, M, cyc, n); break;
    case 2: sweep_wave<2>(S, H, M, cyc, n); break;
#ifdef LOOKAHEAD
    default: pivot_wave(S, cyc, n); break;
#else
    default: sweep_wave<3>(S, H, M, cyc, n); break;
#endif
  }
}

__device__ __forceinline__ void ldl_dinv_row(Smem &S, const int s, const int g, const int c) {
    // LDL' in pivot order: backward stable for the positive definite block (closed-form 2 x 2 determinants lose cond(D) eps)
    const double(*D)[4] = S.Draw;
    const double d00 = D[0][0], d10 = D[1][0], d20 = D[2][0], d30 = D[3][0];
    const double d11 = D[1][1], d21 = D[2][1], d31 = D[3][1], d22 = D[2][2], d32 = D[3][2], d33 = D[3][3];
    const double i0 = rcp_nr(d00);
    const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
    const double e1 = dfma(-l10, d10, d11), i1 = rcp_nr(e1);
    const double m21 = dfma(-l20, d10, d21), m31 = dfma(-l30, d10, d31);
    const double l21 = m21 * i1, l31 = m31 * i1;
    const double e2 = dfma(-l21, m21, dfma(-l20, d20, d22)), i2 = rcp_nr(e2);
    const double m32 = dfma(-l31, m21, dfma(-l30, d20, d32));
    const double l32 = m32 * i2;
    const double e3 = dfma(-l32, m32, dfma(-l31, m31, dfma(-l30, d30, d33))), i3 = rcp_nr(e3);
    double y0 = (g == 0) ? 1.0 : 0.0, y1 = (g == 1) ? 1.0 : 0.0, y2 = (g == 2) ? 1.0 : 0.0, y3 = (g == 3) ? 1.0 : 0.0;
    y1 = dfma(-l10, y0, y1);
    y2 = dfma(-l21, y1, dfma(-l20, y0, y2));
    y3 = dfma(-l32, y2, dfma(-l31, y1, dfma(-l30, y0, y3)));
    const double x3 = y3 * i3;
    const double x2 = dfma(-l32, x3, y2 * i2);
    const double x1 = dfma(-l31, x3, dfma(-l21, x2, y1 * i1));
    const double x0 = dfma(-l30, x3, dfma(-l20, x2, dfma(-l10, x1, y0 * i0)));
    if (c == 0) {
      double *dst = S.Dinv[s & 1][g];
      dst[0] = x0, dst[1] = x1, dst[2] = x2, dst[3] = x3;
    }
  }

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 3072, n = argc > 2 ? atoi(argv[2]) : 120;
  std::vector<double> H((size_t)nb * N * N, 0.0), M((size_t)nb * N * N, 0.0);
  srand(1);
  const int ndist = nb < 8 ? nb : 8;  // distinct matrices (host time); the rest of the batch repeats them
  for (int b = 0; b < ndist; ++b) {   // H = B'B + alpha I, B 40 x 120: rank deficient like B'SB, cond ~ 1e6
    std::vector<double> B(40 * N);
    for (auto &v : B) v = (rand() / (double)RAND_MAX - 0.5) * 8.0;
    double *h = &H[(size_t)b * N * N];
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = (i == j) ? 2e-4 : 0.0;
        for (int k = 0; k < 40; ++k) s += B[k * N + i] * B[k * N + j];
        h[i * N + j] = s;
      }
  }
  for (int b = ndist; b < nb; ++b) memcpy(&H[(size_t)b * N * N], &H[(size_t)(b % ndist) * N * N], sizeof(double) * N * N);
  double *dH, *dM;
  long long *dc;
  hipMalloc(&dH, H.size() * 8), hipMalloc(&dM, M.size() * 8), hipMalloc(&dc, nb * 8);
  hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(sweep_kernel, dim3(nb), dim3(NT), sizeof(Smem), 0, dH, dM, dc, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.3f ms for %d matrices (incl. the global load/store of the matrices)\n", rep, ms, nb);
  }
  if (hipGetLastError() != hipSuccess) return printf("launch failed\n"), 1;
  hipMemcpy(M.data(), dM, M.size() * 8, hipMemcpyDeviceToHost);
  std::vector<long long> cyc(nb);
  hipMemcpy(cyc.data(), dc, nb * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : cyc) mean += v;
  printf("sweep cycles per workgroup (smem %zu B, 3 per CU): mean %.0f\n", sizeof(Smem), mean / nb);
  // accuracy: |H M - I|_max and symmetry on the first matrices
  for (int b = 0; b < 2; ++b) {
    const double *h = &H[(size_t)b * N * N], *m = &M[(size_t)b * N * N];
    double worst = 0, scale = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += h[i * N + k] * m[k * N + j];
        worst = fmax(worst, fabs(s - (i == j)));
        scale = fmax(scale, fabs(m[i * N + j]));
      }
    printf("matrix %d: |H M - I|_max = %.3e, |M|_max = %.3e\n", b, worst, scale);
  }
  return 0;
}
