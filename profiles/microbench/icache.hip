// icache.hip -- diagnostic: does a long straight-line fp64 kernel (like k_linearize) run out of instruction
// fetch bandwidth when several waves per SIMD execute it?  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int N>
__global__ void straight(double *out, unsigned long long *cyc, double seed) {
  double x0 = seed + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, b = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  STAMP(t0);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
  }
  STAMP(t1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int N>
__global__ void looped(double *out, unsigned long long *cyc, double seed, int n) {
  double x0 = seed + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, b = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int i = 0; i < n; ++i) {
    x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
  }
  STAMP(t1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3;
}

int main() {
  constexpr int N = 4096;
  double *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 8 * 64 * 8192); (void)hipMalloc(&cyc, 8 * 8192);
  std::vector<unsigned long long> h(8192);
  for (int blocks : {256, 1024, 2048, 3072, 4096}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(straight<N>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.25);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
      hipLaunchKernelGGL(looped<N>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.25, N / 4);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
      double l = 0; for (int i = 0; i < blocks; ++i) l += h[i];
      printf("blocks %5d rep %d  straight-line %6.2f cycles/op   loop %6.2f cycles/op\n", blocks, rep, s / blocks / N, l / blocks / N);
    }
  }
  return 0;
}
