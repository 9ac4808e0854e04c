// Diagnostic (round 5): how accurate is v_rcp_f64 on gfx950, and what do Newton steps and the quotient's correction step leave?
//   r0 = v_rcp_f64(b);  r1 = one Newton step;  r2 = two;  q(a, r) = a r corrected once: q + r fma(-b, q, a)
// Prints the largest relative error of r0, r1, r2 against 1 / b and the number of quotients q(a, r1), q(a, r2) that differ from the correctly
// rounded a / b, over 2^22 random operands per magnitude class.   hipcc --offload-arch=gfx950 -O2 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const double *a, const double *b, double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double bb = b[i], aa = a[i];
  const double r0 = __builtin_amdgcn_rcp(bb);
  const double r1 = __builtin_fma(__builtin_fma(-bb, r0, 1.0), r0, r0);
  const double r2 = __builtin_fma(__builtin_fma(-bb, r1, 1.0), r1, r1);
  const double q1 = aa * r1, q2 = aa * r2;
  out[5 * i + 0] = r0; out[5 * i + 1] = r1; out[5 * i + 2] = r2;
  out[5 * i + 3] = __builtin_fma(__builtin_fma(-bb, q1, aa), r1, q1);
  out[5 * i + 4] = __builtin_fma(__builtin_fma(-bb, q2, aa), r2, q2);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> a(n), b(n), out(5 * (size_t)n);
  double *da, *db, *dout;
  if (hipMalloc(&da, n * 8) != hipSuccess || hipMalloc(&db, n * 8) != hipSuccess || hipMalloc(&dout, 5 * (size_t)n * 8) != hipSuccess) return 1;
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (double scale : {1.0, 1e-6, 1e6, 1e-150, 1e150}) {
    for (int i = 0; i < n; ++i) { a[i] = (rnd() - 0.5) * 4.0; b[i] = (0.5 + rnd()) * scale * (rnd() < 0.5 ? -1.0 : 1.0); }
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dout, n);
    (void)hipMemcpy(out.data(), dout, 5 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
      const long double t = 1.0L / (long double)b[i];
      e0 = fmax(e0, (double)fabsl((out[5 * i] - t) / t)); e1 = fmax(e1, (double)fabsl((out[5 * i + 1] - t) / t)); e2 = fmax(e2, (double)fabsl((out[5 * i + 2] - t) / t));
      const double q = a[i] / b[i];
      bad1 += out[5 * i + 3] != q; bad2 += out[5 * i + 4] != q;
    }
    printf("|b| ~ %-7g: max rel error of v_rcp_f64 %.2e, after one Newton step %.2e, after two %.2e; corrected quotients that differ from a / b: %ld (one step), %ld (two steps) of %d\n",
           scale, e0, e1, e2, bad1, bad2, n);
  }
  return 0;
}
