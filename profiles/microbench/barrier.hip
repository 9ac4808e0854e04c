// barrier.hip -- diagnostic: cost of one producer->consumer hop between wavefronts of a block through LDS and
// s_barrier, as a function of the number of wavefronts at the barrier (the rollout and backward kernels pay
// one such hop per knot).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int WAVES>
__global__ void hop(double *out, unsigned long long *cyc, int iters) {
  __shared__ double buf[2][WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double x = 1.0 + lane * 1e-9;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int i = 0; i < iters; ++i) {
    buf[i & 1][w][lane] = x;                         // publish
    __syncthreads();
    x = buf[i & 1][(w + 1) % WAVES][lane] * 1.0000001 + 1e-9;  // consume the neighbour's value
  }
  STAMP(t1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = x;
}
template <int WAVES>
void run(int blocks) {
  const int iters = 2000;
  double *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 8 * 64 * WAVES * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(hop<WAVES>, dim3(blocks), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("waves/block %d blocks %4d: %7.1f cycles per write+barrier+read hop\n", WAVES, blocks, s / blocks / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int blocks : {16, 256}) { run<1>(blocks); run<2>(blocks); run<3>(blocks); run<4>(blocks); run<5>(blocks); run<8>(blocks); }
  return 0;
}
