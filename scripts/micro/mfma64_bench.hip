// Developer microbenchmark: issue cost of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on gfx950 (cycles per instruction per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double *out, long long *cyc, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma(double *out, long long *cyc, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, b, acc[i]);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out;
  long long *cyc;
  hipMalloc(&out, 8 << 20);
  hipMalloc(&cyc, 8 << 12);
  long long h[8];
  const int iters = 2000;
  for (int nt : {64, 256, 512}) {
#define RUN(N)                                                                                              \
  hipLaunchKernelGGL(k_mfma<N>, dim3(1), dim3(nt), 0, 0, out, cyc, iters);                                 \
  hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);                                                              \
  printf("mfma_f64_16x16x4  threads/CU=%d  independent accumulators=%d  cycles per instr per wave = %.1f\n", nt, N, \
         (double)h[0] / (iters * N));
    RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
    hipLaunchKernelGGL(k_fma, dim3(1), dim3(nt), 0, 0, out, cyc, iters);
    hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64         threads/CU=%d  cycles per instr per wave = %.1f\n", nt, (double)h[0] / (iters * 16));
  }
  return 0;
}
