// coissue.hip -- diagnostic: do fp64 MFMAs of one wavefront and fp64 VALU instructions of another wavefront on the SAME
// SIMD overlap on gfx950?  (The question behind the saturated-machine bound of k_backward: 7 MFMAs + ~230 VALU instructions
// per knot.)  Blocks of eight wavefronts, one block per CU (LDS hog): wavefronts w and w + 4 share SIMD w % 4.
//   mode 0: w < 4 run independent v_mfma_f64_16x16x4_f64, w >= 4 idle       mode 1: w < 4 idle, w >= 4 independent v_fma_f64
//   mode 2: both at once                                                    mode 3: both halves run MFMAs   mode 4: both FMAs
// Prints shader cycles per instruction for either kind (s_memtime) and the clock the loop ran at (cycles / s_memrealtime).
// Not part of the product.   hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define RSTAMP(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

__global__ __launch_bounds__(512) void k(double *out, unsigned long long *cyc, unsigned long long *real, int iters, int mode, double seed) {
  __shared__ double hog[9000];  // 72 KB: one block per CU (two would need 144 KB + ... of 160)
  hog[threadIdx.x] = seed;
  __syncthreads();
  const int w = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0 && w < 4) || (mode == 2 && w < 4) || mode == 3;
  const bool do_fma = (mode == 1 && w >= 4) || (mode == 2 && w >= 4) || mode == 4;
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  d4 p = {a, a, a, a}, q = {a, a, a, a}, r = {a, a, a, a}, s = {a, a, a, a};
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  __syncthreads();
  RSTAMP(r0);
  STAMP(t0);
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {  // four independent accumulators: 4 MFMAs per iteration
      p = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b, p, 0, 0, 0);
      q = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b, q, 0, 0, 0);
      r = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b, r, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b, s, 0, 0, 0);
    }
  } else if (do_fma) {
    for (int i = 0; i < iters; ++i) {  // 64 independent-enough FMAs per iteration (8 chains, 8 deep)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
        x4 = __builtin_fma(x4, b, c); x5 = __builtin_fma(x5, b, c); x6 = __builtin_fma(x6, b, c); x7 = __builtin_fma(x7, b, c);
      }
    }
  }
  STAMP(t1);
  RSTAMP(r1);
  const int g = blockIdx.x * 8 + w;
  if ((threadIdx.x & 63) == 0) { cyc[g] = t1 - t0; real[g] = r1 - r0; }
  out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p[0] + q[1] + r[2] + s[3] + hog[(threadIdx.x * 7) % 9000];
}

int main() {
  const int blocks = 256, iters = 2000;
  double *out; unsigned long long *cyc, *real;
  hipMalloc(&out, sizeof(double) * blocks * 512); hipMalloc(&cyc, 8 * blocks * 8); hipMalloc(&real, 8 * blocks * 8);
  std::vector<unsigned long long> hc(blocks * 8), hr(blocks * 8);
  const char *names[5] = {"MFMA on w<4, w>=4 idle", "FMA on w>=4, w<4 idle", "MFMA on w<4 + FMA on w>=4 (same SIMDs)", "MFMA on all eight", "FMA on all eight"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, real, iters, mode, 1.0);
    hipDeviceSynchronize();
    hipMemcpy(hc.data(), cyc, 8 * blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), real, 8 * blocks * 8, hipMemcpyDeviceToHost);
    double cm = 0, cf = 0, rm = 0, rf = 0; int nm = 0, nf = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) {
        const bool m = (mode == 0 && w < 4) || (mode == 2 && w < 4) || mode == 3, f = (mode == 1 && w >= 4) || (mode == 2 && w >= 4) || mode == 4;
        if (m) { cm += hc[b * 8 + w]; rm += hr[b * 8 + w]; ++nm; }
        if (f) { cf += hc[b * 8 + w]; rf += hr[b * 8 + w]; ++nf; }
      }
    printf("%-42s", names[mode]);
    if (nm) printf("  MFMA: %6.1f cycles each, clock %.2f GHz", cm / nm / (4.0 * iters), (cm / nm) / (rm / nm) * 0.1);
    if (nf) printf("  FMA: %5.2f cycles each, clock %.2f GHz", cf / nf / (64.0 * iters), (cf / nf) / (rf / nf) * 0.1);
    printf("\n");
  }
  return 0;
}
