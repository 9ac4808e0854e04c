// Microbenchmark (round 4): what a launch boundary between two DEPENDENT kernels costs on the device, and whether a hipGraph makes it
// cheaper.  A chain of N kernels on one stream, each one block of 320 threads that spins for `work` shader cycles (0: returns at once),
// launched (a) one by one with the host running ahead (what the solver's round loop does), (b) as one instantiated hipGraph of the
// same N kernel nodes.  Printed: wall time per kernel of the chain, both ways, for a few amounts of work.
// build: hipcc --offload-arch=gfx950 -O2 -o graph_gap graph_gap.hip      run: ./graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tick(unsigned long long *out, int work) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  do {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  } while ((long long)(t1 - t0) < work);
  if (threadIdx.x == 0) out[blockIdx.x] += 1;  // (the chain's data dependence)
}

// the same with the combined launch's footprint: 62 KB of LDS per block (one block per CU) and a few hundred bytes of kernel arguments
struct Fat { double a[160]; };
__global__ __launch_bounds__(320) void k_tick_fat(unsigned long long *out, int work, Fat f) {
  __shared__ double lds[62 * 128];
  lds[threadIdx.x] = f.a[threadIdx.x & 127];
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  do {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  } while ((long long)(t1 - t0) < work);
  if (threadIdx.x == 0) out[blockIdx.x] += (unsigned long long)lds[(threadIdx.x + 1) & 127];
}

int main() {
  const int N = 200, blocks = 256;
  unsigned long long *d;
  CHECK(hipMalloc(&d, sizeof(*d) * blocks));
  CHECK(hipMemset(d, 0, sizeof(*d) * blocks));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int work : {0, 20000, 240000}) {  // 0, ~8 us, ~100 us at 2.4 GHz
    auto run_stream = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tick, dim3(blocks), dim3(320), 0, s, d, work); };
    run_stream();
    CHECK(hipStreamSynchronize(s));
    double best_s = 1e9, best_g = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      run_stream();
      CHECK(hipStreamSynchronize(s));
      best_s = std::min(best_s, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    run_stream();
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CHECK(hipGraphLaunch(ge, s));
      CHECK(hipStreamSynchronize(s));
      best_g = std::min(best_g, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    printf("work %6d cycles: %7.2f us per kernel launched one by one, %7.2f us as a graph of %d nodes\n", work, best_s, best_g, N);
  }
  {
    Fat f;
    for (int k = 0; k < 160; ++k) f.a[k] = 1.0;
    const int work = 240000;
    auto run_fat = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tick_fat, dim3(blocks), dim3(320), 0, s, d, work, f); };
    run_fat();
    CHECK(hipStreamSynchronize(s));
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      run_fat();
      CHECK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    printf("work %6d cycles, 62 KB of LDS per block and 1.3 KB of arguments: %7.2f us per kernel launched one by one\n", work, best);
  }
  return 0;
}
