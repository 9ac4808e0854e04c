// writebw.hip -- diagnostic: HBM write (and read) bandwidth of plain streaming kernels at the sizes k_linearize
// moves (75 MB per launch at B = 1024) and larger.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void wr(d2 *p, long n, double v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = d2{v, v};
}
__global__ void rd(const d2 *p, long n, double *out) {
  double s = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { d2 x = p[i]; s += x[0] + x[1]; }
  if (s == 1.2345) out[0] = s;
}
int main() {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  double *out; (void)hipMalloc(&out, 8);
  for (long mb : {75L, 300L, 2400L}) {
    const long n = mb * 1000000 / 16;
    d2 *p; (void)hipMalloc(&p, n * 16);
    for (int pass = 0; pass < 2; ++pass) {
      float best = 1e9f;
      for (int r = 0; r < 6; ++r) {
        (void)hipEventRecord(a);
        if (pass == 0) hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, p, n, 1.0 + r);
        else hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, p, n, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      printf("%s %5ld MB: %7.1f us  %6.2f TB/s\n", pass == 0 ? "write" : "read ", mb, best * 1e3, mb * 1e6 / (best * 1e-3) / 1e12);
    }
    (void)hipFree(p);
  }
  return 0;
}
