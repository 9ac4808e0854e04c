// ubench.hip -- diagnostic micro-benchmarks for the fp64 serial chains of the iLQR kernels on gfx950.
// Not part of the product.  Measures, per wave, shader cycles (s_memtime) and wall time
// (s_memrealtime, 100 MHz) of dependent / independent fp64 FMA chains, the fp64 MFMA, v_rcp_f64,
// ds_bpermute and v_readlane, at a low-occupancy grid (16 waves) and a full grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define RSTAMP(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, unsigned long long *real, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  d4 acc = {a, a, a, a};
  d4 acc2 = {a, a, a, a};
  unsigned long long t0, t1, r0, r1;
  RSTAMP(r0);
  STAMP(t0);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // dependent fma chain, 8 per iteration
#pragma unroll
      for (int u = 0; u < 8; ++u) x0 = __builtin_fma(x0, b, c);
    } else if (MODE == 1) {  // 8 independent fma
      x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
      x4 = __builtin_fma(x4, b, c); x5 = __builtin_fma(x5, b, c); x6 = __builtin_fma(x6, b, c); x7 = __builtin_fma(x7, b, c);
    } else if (MODE == 2) {  // dependent mfma f64 chain through C, 4 per iteration
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b, acc, 0, 0, 0);
    } else if (MODE == 3) {  // dependent mfma through B operand (result -> operand), 4 per iteration
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, acc[0], acc2, 0, 0, 0); }
    } else if (MODE == 4) {  // rcp + 2 newton, dependent, 2 per iteration
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        double r = __builtin_amdgcn_rcp(x0);
        r = __builtin_fma(__builtin_fma(-x0, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-x0, r, 1.0), r, r);
        x0 = r + 1.5;
      }
    } else if (MODE == 5) {  // IEEE division, dependent, 2 per iteration
#pragma unroll
      for (int u = 0; u < 2; ++u) x0 = 1.0 / x0 + 1.5;
    } else if (MODE == 6) {  // ds_bpermute f64 (2 dwords) dependent, 4 per iteration
#pragma unroll
      for (int u = 0; u < 4; ++u) x0 = __shfl(x0, (threadIdx.x + 17) & 63) + c;
    } else if (MODE == 7) {  // readlane f64 dependent, 4 per iteration
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        long long v = __double_as_longlong(x0);
        int lo = __builtin_amdgcn_readlane((int)v, 13), hi = __builtin_amdgcn_readlane((int)(v >> 32), 13);
        x0 = __longlong_as_double(((long long)hi << 32) | (unsigned)lo) + threadIdx.x * c;
      }
    } else if (MODE == 8) {  // sqrt dependent, 2 per iteration
#pragma unroll
      for (int u = 0; u < 2; ++u) x0 = sqrt(x0) + 2.0;
    } else if (MODE == 9) {  // sincos dependent, 1 per iteration
      double s, co;
      sincos(x0, &s, &co);
      x0 = s + co * 0.5;
    } else if (MODE == 10) {  // atan2 dependent, 1 per iteration
      x0 = atan2(x0, b) + 0.3;
    } else if (MODE == 11) {  // 8 independent mul+add pairs of f32-free fp64 add
      x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
    }
  }
  STAMP(t1);
  RSTAMP(r1);
  if (threadIdx.x == 0) {
    cyc[blockIdx.x] = t1 - t0;
    real[blockIdx.x] = r1 - r0;
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char *name, int per_iter, int blocks) {
  const int iters = 2000;
  double *out;
  unsigned long long *cyc, *real;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, 8 * blocks);
  hipMalloc(&real, 8 * blocks);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, real, iters, 1.25);
  hipDeviceSynchronize();
  std::vector<unsigned long long> hc(blocks), hr(blocks);
  hipMemcpy(hc.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), real, 8 * blocks, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < blocks; ++i) { c += hc[i]; r += hr[i]; }
  c /= blocks; r /= blocks;
  printf("%-28s blocks %5d  cycles/op %8.2f  ns/op %8.2f  clock %.3f GHz\n", name, blocks, c / (iters * per_iter),
         r * 10.0 / (iters * per_iter), c / (r * 10.0));
  hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
  for (int blocks : {16, 1024, 2048, 3072, 4096, 8192}) {
    run<0>("fma_f64 dependent", 8, blocks);
    run<1>("fma_f64 8 independent", 8, blocks);
    run<11>("add_f64 8 independent", 8, blocks);
    run<2>("mfma_f64 dep via C", 4, blocks);
    run<3>("mfma_f64 dep via B", 4, blocks);
    run<4>("rcp+2NR dependent", 2, blocks);
    run<5>("ieee div dependent", 2, blocks);
    run<6>("shfl f64 dependent", 4, blocks);
    run<7>("readlane f64 dependent", 4, blocks);
    run<8>("sqrt dependent", 2, blocks);
    run<9>("sincos dependent", 1, blocks);
    run<10>("atan2 dependent", 1, blocks);
  }
  return 0;
}
