// Developer check: relative error of v_rcp_f64 (raw, and after one / two Newton steps) against IEEE division.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r1[i] = r;
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    double m = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);  // [1,2)
    int e = (int)((s >> 3) % 41) - 20;
    x[i] = ldexp(m, e);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8), hipMalloc(&d0, n * 8), hipMalloc(&d1, n * 8), hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  std::vector<double> r0(n), r1(n), r2(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    long double ex = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl(((long double)r0[i] - ex) / ex));
    e1 = fmax(e1, (double)fabsl(((long double)r1[i] - ex) / ex));
    e2 = fmax(e2, (double)fabsl(((long double)r2[i] - ex) / ex));
  }
  printf("max relative error of v_rcp_f64: raw %.3e (2^%.1f), one Newton step %.3e, two %.3e\n", e0, log2(e0), e1, e2);
  return 0;
}
