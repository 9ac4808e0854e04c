// execmask.hip -- diagnostic: does a fp64 VALU instruction issue faster when only 16 or 32 lanes of the
// wavefront are active?  (If it did, a rollout with 16 trajectories per wavefront would pay.)  One wave,
// 8 independent / 8 dependent FMAs per iteration, with the first `active` lanes running the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int DEP>
__global__ void k(double *out, unsigned long long *cyc, int iters, int active, double seed) {
  double b = 1.0000001, c = 1e-9;
  double x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < active) {
    STAMP(t0);
    for (int i = 0; i < iters; ++i) {
      if (DEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u) x0 = __builtin_fma(x0, b, c);
      } else {
        x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
        x4 = __builtin_fma(x4, b, c); x5 = __builtin_fma(x5, b, c); x6 = __builtin_fma(x6, b, c); x7 = __builtin_fma(x7, b, c);
      }
    }
    STAMP(t1);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  }
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  double *out; unsigned long long *cyc, h;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int dep = 0; dep < 2; ++dep)
    for (int active : {64, 32, 16, 1}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (dep) k<1><<<1, 64>>>(out, cyc, iters, active, 1.0); else k<0><<<1, 64>>>(out, cyc, iters, active, 1.0);
        hipDeviceSynchronize();
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s fma, %2d active lanes: %.2f cycles (s_memtime units) per instruction\n", dep ? "dependent  " : "independent", active, (double)h / (iters * 8.0));
    }
  return 0;
}
