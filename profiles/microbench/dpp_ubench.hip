// dpp_ubench.hip -- diagnostic micro-benchmark (not part of the product): what one wavefront alone on its SIMD pays
// for the primitives of a rollout that gives 16 lanes to a trajectory -- fp64 multiply-add with a DPP row broadcast
// operand (v_fmac_f64_dpp row_newbcast, the only DPP form fp64 arithmetic has on gfx950), the 64-bit row broadcast
// move, a 64-bit lane permutation as two v_mov_b32_dpp quad_perm, and LDS operand reads -- in dependent and
// independent streams.  Cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int R>
__device__ __forceinline__ double fmac_bc(double acc, double src, double m) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(R));
  return acc;
}
template <int R>
__device__ __forceinline__ double bc(double x) {
  return __builtin_amdgcn_mov_dpp(x, 0x150 + R, 0xf, 0xf, false);
}
// rotate inside every quad: lane 4k+i <- lane 4k+(i+1)%3 for i < 3, lane 4k+3 stays (quad_perm:[1,2,0,3] = 0xC9)
__device__ __forceinline__ double rot1(double x) {
  const long long v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)v, 0xC9, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), 0xC9, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double rot1_mov(double x) {  // mov_dpp (no `old` operand)
  const long long v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_mov_dpp((int)v, 0xC9, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), 0xC9, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int iters, double seed) {
  __shared__ double lds[64 * 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 16; i += 64) lds[i] = seed + i * 1e-3;
  __syncthreads();
  double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  double m0 = b, m1 = b * 1.1, m2 = b * 1.2, m3 = b * 1.3;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 dependent fmac_dpp (one accumulator)
      x0 = fmac_bc<0>(x0, x1, m0); x0 = fmac_bc<1>(x0, x1, m1); x0 = fmac_bc<2>(x0, x1, m2); x0 = fmac_bc<3>(x0, x1, m3);
      x0 = fmac_bc<4>(x0, x1, m0); x0 = fmac_bc<5>(x0, x1, m1); x0 = fmac_bc<6>(x0, x1, m2); x0 = fmac_bc<7>(x0, x1, m3);
      x0 *= 1e-3;
    } else if (MODE == 1) {  // 8 fmac_dpp into 4 accumulators (two rounds)
      x0 = fmac_bc<0>(x0, x7, m0); x1 = fmac_bc<1>(x1, x7, m1); x2 = fmac_bc<2>(x2, x7, m2); x3 = fmac_bc<3>(x3, x7, m3);
      x0 = fmac_bc<4>(x0, x7, m0); x1 = fmac_bc<5>(x1, x7, m1); x2 = fmac_bc<6>(x2, x7, m2); x3 = fmac_bc<7>(x3, x7, m3);
      x0 *= 1e-3; x1 *= 1e-3; x2 *= 1e-3; x3 *= 1e-3;
    } else if (MODE == 2) {  // 8 x (mov_b64_dpp broadcast + fma), 4 accumulators
      x0 = __builtin_fma(bc<0>(x7), m0, x0); x1 = __builtin_fma(bc<1>(x7), m1, x1); x2 = __builtin_fma(bc<2>(x7), m2, x2); x3 = __builtin_fma(bc<3>(x7), m3, x3);
      x0 = __builtin_fma(bc<4>(x7), m0, x0); x1 = __builtin_fma(bc<5>(x7), m1, x1); x2 = __builtin_fma(bc<6>(x7), m2, x2); x3 = __builtin_fma(bc<7>(x7), m3, x3);
      x0 *= 1e-3; x1 *= 1e-3; x2 *= 1e-3; x3 *= 1e-3;
    } else if (MODE == 3) {  // 8 independent 64-bit quad rotations (update_dpp) + add to keep them
      x0 += rot1(x4); x1 += rot1(x5); x2 += rot1(x6); x3 += rot1(x7);
      x4 += rot1(x0); x5 += rot1(x1); x6 += rot1(x2); x7 += rot1(x3);
      x0 *= 0.1; x1 *= 0.1; x2 *= 0.1; x3 *= 0.1; x4 *= 0.1; x5 *= 0.1; x6 *= 0.1; x7 *= 0.1;
    } else if (MODE == 4) {  // the same with mov_dpp
      x0 += rot1_mov(x4); x1 += rot1_mov(x5); x2 += rot1_mov(x6); x3 += rot1_mov(x7);
      x4 += rot1_mov(x0); x5 += rot1_mov(x1); x6 += rot1_mov(x2); x7 += rot1_mov(x3);
      x0 *= 0.1; x1 *= 0.1; x2 *= 0.1; x3 *= 0.1; x4 *= 0.1; x5 *= 0.1; x6 *= 0.1; x7 *= 0.1;
    } else if (MODE == 5) {  // reference: 16 independent fma (the adds and muls of modes 3, 4 are 16 such)
      x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
      x4 = __builtin_fma(x4, b, c); x5 = __builtin_fma(x5, b, c); x6 = __builtin_fma(x6, b, c); x7 = __builtin_fma(x7, b, c);
      x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
      x4 = __builtin_fma(x4, b, c); x5 = __builtin_fma(x5, b, c); x6 = __builtin_fma(x6, b, c); x7 = __builtin_fma(x7, b, c);
    } else if (MODE == 6) {  // 8 LDS reads of 8 bytes (fixed offsets from one base), then 8 fma that use them
      const double *p = lds + lane;
      const double l0 = p[0], l1 = p[64], l2 = p[128], l3 = p[192], l4 = p[256], l5 = p[320], l6 = p[384], l7 = p[448];
      x0 = __builtin_fma(x0, b, l0); x1 = __builtin_fma(x1, b, l1); x2 = __builtin_fma(x2, b, l2); x3 = __builtin_fma(x3, b, l3);
      x4 = __builtin_fma(x4, b, l4); x5 = __builtin_fma(x5, b, l5); x6 = __builtin_fma(x6, b, l6); x7 = __builtin_fma(x7, b, l7);
      x0 *= 0.1; x1 *= 0.1; x2 *= 0.1; x3 *= 0.1; x4 *= 0.1; x5 *= 0.1; x6 *= 0.1; x7 *= 0.1;
      asm volatile("" ::: "memory");
    } else if (MODE == 7) {  // cross product by rotations: w = a x b with a = (x0), b = (x1) held 3 lanes per quad
      const double a1 = rot1_mov(x0), a2 = rot1_mov(a1), b1 = rot1_mov(x1), b2 = rot1_mov(b1);
      const double w = a1 * b2 - a2 * b1;
      const double w1 = rot1_mov(w), w2 = rot1_mov(w1);
      const double v = a1 * w2 - a2 * w1;
      x1 = __builtin_fma(v, 1e-3, x1 * 0.5);
    } else if (MODE == 8) {  // dependent fma chain, 8 (reference)
#pragma unroll
      for (int u = 0; u < 8; ++u) x0 = __builtin_fma(x0, b, c);
    }
  }
  STAMP(t1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name, int per_iter, int blocks) {
  const int iters = 2000;
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, 8 * blocks);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.25);
  hipDeviceSynchronize();
  std::vector<unsigned long long> hc(blocks);
  hipMemcpy(hc.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < blocks; ++i) c += hc[i];
  c /= blocks;
  printf("%-58s blocks %4d  cycles/iteration %8.1f  per listed op %7.2f\n", name, blocks, c / iters, c / (iters * per_iter));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {16, 256}) {
    run<8>("fma_f64 x8 dependent", 8, blocks);
    run<5>("fma_f64 x16 independent", 16, blocks);
    run<0>("fmac_f64_dpp x8 dependent (+1 mul)", 9, blocks);
    run<1>("fmac_f64_dpp x8 on 4 accumulators (+4 mul)", 12, blocks);
    run<2>("(mov_b64_dpp + fma) x8 on 4 accumulators (+4 mul)", 20, blocks);
    run<3>("64-bit quad rotation (2 update_dpp) x8 (+8 add +8 mul)", 32, blocks);
    run<4>("64-bit quad rotation (2 mov_dpp) x8 (+8 add +8 mul)", 32, blocks);
    run<6>("ds_read_b64 x8 + 8 fma + 8 mul", 24, blocks);
    run<7>("double cross product by rotations (12 mov, 6 arith)", 18, blocks);
  }
  return 0;
}
