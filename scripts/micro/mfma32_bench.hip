// Developer microbenchmark: issue cost of the fp32 MFMA shapes on gfx950 (cycles per instruction per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(float *out, long long *cyc, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k32(float *out, long long *cyc, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float *out;
  long long *cyc;
  hipMalloc(&out, 8 << 20);
  hipMalloc(&cyc, 8 << 12);
  long long h[8];
  const int iters = 2000;
  for (int nt : {64, 256, 512}) {
#define RUN(K, N, name)                                                                                        \
  hipLaunchKernelGGL(K<N>, dim3(1), dim3(nt), 0, 0, out, cyc, iters);                                          \
  hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);                                                                 \
  printf("%s threads/CU=%d accumulators=%d cycles/instr/wave = %.1f\n", name, nt, N, (double)h[0] / (iters * N));
    RUN(k16, 1, "mfma_f32_16x16x4") RUN(k16, 4, "mfma_f32_16x16x4") RUN(k32, 1, "mfma_f32_32x32x2") RUN(k32, 4, "mfma_f32_32x32x2")
  }
  return 0;
}
