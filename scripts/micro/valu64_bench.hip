// Developer microbenchmark: issue cost (s_memtime ticks per wave instruction) of fp64 VALU ops on gfx950 under
// different operand patterns, against fp32 FMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 16
template <int MODE>
__global__ void k(double *out, long long *cyc, int iters) {
  double acc[N], a[N], b[N];
  float facc[N];
  for (int i = 0; i < N; ++i) acc[i] = i, a[i] = 1.0 + 1e-9 * (threadIdx.x + i), b[i] = 1e-7 * (i + 1), facc[i] = i;
  const double a0 = a[0], b0 = b[0];
  const float fa = (float)a0, fb = (float)b0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (MODE == 0) acc[i] = __builtin_fma(a0, b0, acc[i]);          // shared multiplicands
      if (MODE == 1) acc[i] = __builtin_fma(a[i], b[i], acc[i]);      // all operands distinct
      if (MODE == 2) acc[i] = __builtin_fma(a[i], b0, acc[i]);        // one shared (the sweep's pattern: q_i * p_j)
      if (MODE == 3) acc[i] = acc[i] * a0;                            // mul
      if (MODE == 4) acc[i] = acc[i] + b0;                            // add
      if (MODE == 5) facc[i] = __builtin_fmaf(fa, fb, facc[i]);       // fp32 fma
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < N; ++i) s += acc[i] + facc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;  // one entry per wave
}
int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 8 << 20);
  hipMalloc(&cyc, 8 << 12);
  long long h[16];
  const int iters = 4000;
  const char *names[6] = {"fma_f64 shared a,b", "fma_f64 distinct", "fma_f64 one shared", "mul_f64", "add_f64", "fma_f32"};
  for (int nt : {256, 512, 768, 1024}) {
#define RUN(M)                                                                                  \
  hipLaunchKernelGGL(k<M>, dim3(1), dim3(nt), 0, 0, out, cyc, iters);                            \
  hipMemcpy(h, cyc, 8 * (nt / 64), hipMemcpyDeviceToHost);                                       \
  {                                                                                              \
    long long mx = 0, mn = h[0];                                                                 \
    for (int w = 0; w < nt / 64; ++w) mx = h[w] > mx ? h[w] : mx, mn = h[w] < mn ? h[w] : mn;    \
    printf("%-22s waves/SIMD=%d  ticks per wave instruction: slowest wave %.2f, fastest %.2f -> SIMD rate %.2f ticks/instr\n", \
           names[M], nt / 256 ? nt / 256 : 1, (double)mx / (iters * N), (double)mn / (iters * N),  \
           (double)mx / (iters * N) / (nt >= 256 ? nt / 256 : 1));                               \
  }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  }
  return 0;
}
