// Diagnostic (round 4): what a v_mfma_f64_16x16x4_f64 costs a wavefront, by how it depends on the one before -- chained through the
// accumulator (srcC = the previous result), through the A operand, through the B operand, independent, and in the pattern of the
// backward pass's knot (T = V M: three chained through C, A from the previous tile; H = C + M^T T: three chained through C, B
// from T) -- with 1, 2 or 4 wavefronts on a SIMD.  One kernel instantiation per pattern (no branches in the timed loop), 24 or
// more MFMAs per iteration.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_chain profiles/microbench/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define MF4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
template <int MODE>
__global__ void k(int iters, double *out, long long *cyc, long long *rt) {
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  d4 acc = {0, 0, 0, 0}, r0 = acc, r1 = acc, r2 = acc, r3 = acc, r4 = acc, r5 = acc;
  double va[3] = {a, a + 1, a + 2}, m[3] = {b, b + 1, b + 2};
  long long rbeg = wall_clock64(), c0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (MODE == 0) {  // six chained through the accumulator
        acc = MF(a, b, acc); acc = MF(a, b, acc); acc = MF(a, b, acc); acc = MF(a, b, acc); acc = MF(a, b, acc); acc = MF(a, b, acc);
      } else if constexpr (MODE == 1) {  // chained through A
        acc = MF(acc[0], b, r0); acc = MF(acc[1], b, r0); acc = MF(acc[0], b, r0); acc = MF(acc[1], b, r0); acc = MF(acc[0], b, r0); acc = MF(acc[1], b, r0);
      } else if constexpr (MODE == 2) {  // chained through B
        acc = MF(a, acc[0], r0); acc = MF(a, acc[1], r0); acc = MF(a, acc[0], r0); acc = MF(a, acc[1], r0); acc = MF(a, acc[0], r0); acc = MF(a, acc[1], r0);
      } else if constexpr (MODE == 3) {  // six independent accumulators
        r0 = MF(a, b, r0); r1 = MF(b, a, r1); r2 = MF(a, a, r2); r3 = MF(b, b, r3); r4 = MF(a, b, r4); r5 = MF(b, a, r5);
      } else if constexpr (MODE == 4) {  // two independent accumulators, alternating
        r0 = MF(a, b, r0); r1 = MF(b, a, r1); r0 = MF(a, b, r0); r1 = MF(b, a, r1); r0 = MF(a, b, r0); r1 = MF(b, a, r1);
      } else if constexpr (MODE == 6) {  // v_mfma_f64_4x4x4_4b_f64 (four 4x4 blocks, 4 passes): chained through the accumulator
        double c4 = acc[0];
        c4 = MF4(a, b, c4); c4 = MF4(a, b, c4); c4 = MF4(a, b, c4); c4 = MF4(a, b, c4); c4 = MF4(a, b, c4); c4 = MF4(a, b, c4);
        acc[0] = c4;
      } else if constexpr (MODE == 7) {  // 4x4x4 chained through B (the T -> H hand-over of a knot re-tiled into 4x4 blocks)
        double c4 = acc[0];
        c4 = MF4(a, c4, r0[0]); c4 = MF4(a, c4, r0[0]); c4 = MF4(a, c4, r0[0]); c4 = MF4(a, c4, r0[0]); c4 = MF4(a, c4, r0[0]); c4 = MF4(a, c4, r0[0]);
        acc[0] = c4;
      } else if constexpr (MODE == 8) {  // 4x4x4, six independent accumulators
        r0[0] = MF4(a, b, r0[0]); r1[0] = MF4(b, a, r1[0]); r2[0] = MF4(a, a, r2[0]); r3[0] = MF4(b, b, r3[0]); r4[0] = MF4(a, b, r4[0]); r5[0] = MF4(b, a, r5[0]);
      } else {  // the knot
        d4 T = {0, 0, 0, 0};
        T = MF(va[0], m[0], T); T = MF(va[1], m[1], T); T = MF(va[2], m[2], T);
        d4 H = {a, b, a, b};
        H = MF(m[0], T[0], H); H = MF(m[1], T[1], H); H = MF(m[2], T[2], H);
        va[0] = H[0]; va[1] = H[1]; va[2] = H[2];
        acc = H;
      }
    }
  }
  long long c1 = clock64(), rend = wall_clock64();
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + r0[0] + r1[1] + r2[2] + r3[3] + r4[0] + r5[1];
  if (threadIdx.x == 0) { cyc[0] = c1 - c0; rt[0] = rend - rbeg; }
}
template <int MODE>
void run(const char *name, int tpb, double *out, long long *cyc, long long *rt) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<1, tpb>>>(iters, out, cyc, rt);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc, hr; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hr, rt, 8, hipMemcpyDeviceToHost);
    // Two clocks: wavefront 0's own loop on the shader clock (clock64 around ITS loop), and the whole launch (HIP events).  With several
    // MFMA-bound wavefronts on a SIMD the oldest one issues its chain back to back and the others wait their turn: wavefront 0's loop takes
    // what it takes alone while the launch takes `waves` times as long.  The per-SIMD rate therefore comes from the LAUNCH's duration
    // (round 4 printed wavefront 0's cycles divided by the waves per SIMD there: "one MFMA every 16 cycles" at four waves -- four times the
    // matrix pipe's peak).
    const int waves = tpb / 256 ? tpb / 256 : 1;
    const double ghz = (double)hc / ((double)hr * 10.0), ns_launch = ms * 1e6 / (24.0 * iters);
    if (rep) printf("%d wave%s per SIMD  %-44s: wavefront 0's own loop %6.1f shader cycles per MFMA; the launch %6.1f ns per MFMA and wave = per SIMD one MFMA every %6.1f cycles (%.2f GHz)\n",
                    waves, tpb > 256 ? "s" : " ", name, (double)hc / (24.0 * iters), ns_launch, ns_launch * ghz / waves, ghz);
  }
}
int main() {
  double *out; long long *cyc, *rt;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8); hipMalloc(&rt, 8);
  const int tpbs[] = {64, 256, 512, 1024};
  for (int w = 0; w < 4; ++w) {
    const int tpb = tpbs[w];
    if (tpb == 64) printf("(one wavefront on the whole GPU)\n");
    run<0>("chained through C (accumulate)", tpb, out, cyc, rt);
    run<1>("chained through A", tpb, out, cyc, rt);
    run<2>("chained through B", tpb, out, cyc, rt);
    run<3>("six independent accumulators", tpb, out, cyc, rt);
    run<4>("two independent accumulators, alternating", tpb, out, cyc, rt);
    run<5>("knot pattern: T (A from last H), H (B from T)", tpb, out, cyc, rt);
    // (the 4 x 4 x 4 instruction -- four independent 4 x 4 blocks per wavefront, 256 flop against 2048: what a dependent one costs)
    run<6>("4x4x4_4b chained through C", tpb, out, cyc, rt);
    run<7>("4x4x4_4b chained through B", tpb, out, cyc, rt);
    run<8>("4x4x4_4b six independent accumulators", tpb, out, cyc, rt);
  }
  return 0;
}
