// Diagnostic (round 4): how many instructions of its OWN a wavefront can issue in the shadow of a chained v_mfma_f64_16x16x4_f64
// (each product waits 64 cycles for its predecessor: mfma_chain.hip) -- K independent instructions between two products of an
// accumulate chain, of four kinds: 32-bit integer VALU, fp64 VALU (v_fma_f64: the same double-precision units as the matrix
// instruction), LDS reads, v_mov_b64_dpp.
// hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_shadow profiles/microbench/mfma_shadow.hip && /tmp/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
template <int KIND, int K>
__global__ void k(int iters, double *out, long long *cyc) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x & 63;
  lds[lane] = lane; lds[lane + 64] = 1.0;
  __syncthreads();
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  d4 acc = {0, 0, 0, 0};
  int x[16]; double y[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { x[q] = lane + q; y[q] = a + q; }
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = MF(a, b, acc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if constexpr (KIND == 0) x[q & 15] = x[q & 15] * 3 + u;                           // integer VALU (v_mad / v_mul_lo ...)
        else if constexpr (KIND == 1) y[q & 15] = __builtin_fma(y[q & 15], b, a);          // fp64 VALU
        else if constexpr (KIND == 2) y[q & 15] += lds[(lane + 8 * q + u) & 1023];         // LDS read (+ an fp64 add)
        else y[q & 15] = __builtin_amdgcn_mov_dpp(y[(q + 1) & 15], 0x150 + 3, 0xf, 0xf, false);  // v_mov_b64_dpp
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long c1 = clock64();
  double s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int q = 0; q < 16; ++q) s += x[q] + y[q];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <int KIND, int K>
void run(const char *name, double *out, long long *cyc) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    k<KIND, K><<<1, 64>>>(iters, out, cyc);
    hipDeviceSynchronize();
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("%-12s K = %2d between products: %7.1f cycles per product (+%6.1f over the bare chain's 64 -> %5.1f per instruction)\n", name, K,
                    (double)hc / (8.0 * iters), (double)hc / (8.0 * iters) - 64.0, K ? ((double)hc / (8.0 * iters) - 64.0) / K : 0.0);
  }
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
  run<0, 0>("none", out, cyc);
  run<0, 4>("int VALU", out, cyc); run<0, 8>("int VALU", out, cyc); run<0, 12>("int VALU", out, cyc); run<0, 16>("int VALU", out, cyc); run<0, 24>("int VALU", out, cyc);
  run<1, 2>("fp64 VALU", out, cyc); run<1, 4>("fp64 VALU", out, cyc); run<1, 8>("fp64 VALU", out, cyc); run<1, 12>("fp64 VALU", out, cyc); run<1, 16>("fp64 VALU", out, cyc);
  run<2, 2>("LDS read", out, cyc); run<2, 4>("LDS read", out, cyc); run<2, 8>("LDS read", out, cyc);
  run<3, 4>("mov_b64_dpp", out, cyc); run<3, 8>("mov_b64_dpp", out, cyc); run<3, 12>("mov_b64_dpp", out, cyc);
  return 0;
}
