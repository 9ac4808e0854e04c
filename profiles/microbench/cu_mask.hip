// Diagnostic (round 6): do CU-masked streams (hipExtStreamCreateWithCUMask) partition the MI355X?  A compute-bound kernel (fixed work per block,
// many blocks) on: the unmasked stream; streams whose 256-bit mask has the low half, the even bits, one bit in four, three bits in four, the
// low 16 / 24 of every 32; and two streams with complementary masks at the same time.  Wall-clock timing around stream synchronisation,
// every line flushed, no stream is destroyed (run under a timeout).
// build + run: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/cu_mask profiles/microbench/cu_mask.hip && timeout 60 /tmp/cu_mask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define SAY(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)
__global__ void burn(int iters, double *out) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
  for (int i = 0; i < iters; ++i) {
    a = __builtin_fma(a, b, 1e-9);
    a = __builtin_fma(a, b, 1e-9);
    a = __builtin_fma(a, b, 1e-9);
    a = __builtin_fma(a, b, 1e-9);
  }
  if (a == 123.0) out[0] = a;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double run(hipStream_t s, int blocks, int iters, double *out) {
  hipStreamSynchronize(s);
  const double t0 = now();
  burn<<<blocks, 256, 0, s>>>(iters, out);
  hipStreamSynchronize(s);
  return now() - t0;
}
int main() {
  const int blocks = 16384, iters = 4000;
  double *out; hipMalloc(&out, 8);
  hipStream_t plain; hipStreamCreate(&plain);
  run(plain, blocks, iters, out);
  const double t_full = run(plain, blocks, iters, out);
  SAY("unmasked stream: %.3f ms\n", t_full);
  struct Pat { const char *name; unsigned word; } pats[] = {{"low 128 of 256 bits", 0}, {"even bits", 0x55555555u}, {"one bit in four", 0x11111111u},
                                                            {"three bits in four", 0x77777777u}, {"low 16 of every 32", 0x0000ffffu}, {"low 24 of every 32", 0x00ffffffu}};
  for (auto &p : pats) {
    unsigned mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = p.word ? p.word : (w < 4 ? 0xffffffffu : 0u);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { SAY("%s: hipExtStreamCreateWithCUMask failed: %s\n", p.name, hipGetErrorString(e)); continue; }
    run(s, blocks, iters, out);
    const double t = run(s, blocks, iters, out);
    SAY("%-22s: %.3f ms = %.2f x the unmasked time\n", p.name, t, t / t_full);
  }
  unsigned ma[8], mb[8];
  for (int w = 0; w < 8; ++w) { ma[w] = 0x77777777u; mb[w] = 0x88888888u; }
  hipStream_t sa, sb;
  if (hipExtStreamCreateWithCUMask(&sa, 8, ma) == hipSuccess && hipExtStreamCreateWithCUMask(&sb, 8, mb) == hipSuccess) {
    run(sa, blocks, iters, out); run(sb, blocks / 4, iters, out);
    hipDeviceSynchronize();
    const double t0 = now();
    burn<<<blocks, 256, 0, sa>>>(iters, out);
    burn<<<blocks / 4, 256, 0, sb>>>(iters, out);
    hipStreamSynchronize(sb);
    const double tb = now() - t0;
    hipStreamSynchronize(sa);
    const double ta = now() - t0;
    SAY("together: %d blocks on three bits in four %.3f ms, %d blocks on the fourth %.3f ms (alone, unmasked, the %d blocks take %.3f ms)\n", blocks, ta, blocks / 4, tb,
        blocks + blocks / 4, t_full * 1.25);
    // and two UNMASKED streams for comparison
    hipStream_t pa, pb; hipStreamCreate(&pa); hipStreamCreate(&pb);
    hipDeviceSynchronize();
    const double t1 = now();
    burn<<<blocks, 256, 0, pa>>>(iters, out);
    burn<<<blocks / 4, 256, 0, pb>>>(iters, out);
    hipStreamSynchronize(pb);
    const double ub = now() - t1;
    hipStreamSynchronize(pa);
    SAY("two unmasked streams: %.3f ms / %.3f ms\n", now() - t1, ub);
  }
  return 0;
}
