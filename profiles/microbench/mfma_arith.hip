// Diagnostic (round 6): WHICH arithmetic does v_mfma_f64_16x16x4_f64 perform?  D = A(16x4) B(4x16) + C over the contraction index k = 0..3.
// Candidates for one element d = c + a0 b0 + a1 b1 + a2 b2 + a3 b3, each compared bit for bit with the instruction's result on random
// operands (including cancelling ones, where the candidates part):
//   seq      fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c))))      four fused multiply-adds in the order of k, the accumulator first
//   rev      the same from k = 3 down to 0
//   pair     c + ((a0 b0 + a1 b1) + (a2 b2 + a3 b3)) with fused products
//   exact    the exactly rounded sum (one rounding): checked on the host with long double where that is exact enough to tell
// Why it matters: if the instruction is `seq`, a gradient wavefront can form Q_uu = C_uu + J_u^T V_xx J_u with plain multiply-adds and get
// the very bits of rows 12..15 of the matrix wavefront's accumulator tile -- the factorisation of Q_uu can then leave the matrix wavefronts
// (k_backward4, six-wavefront form) without changing a bit of the results.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_arith profiles/microbench/mfma_arith.hip && /tmp/mfma_arith
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, double *D, int tiles) {
  // one wavefront per tile: lane l = (j = l & 15, kk = l >> 4) supplies A[j][kk], B[kk][j], C[4 r + kk][j]; receives D[4 r + kk][j]
  const int t = blockIdx.x, lane = threadIdx.x, j = lane & 15, kk = lane >> 4;
  if (t >= tiles) return;
  const double a = A[t * 64 + j * 4 + kk], b = B[t * 64 + kk * 16 + j];
  d4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * r + kk) * 16 + j];
  const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[t * 256 + (4 * r + kk) * 16 + j] = d[r];
}
int main() {
  const int tiles = 4096;
  std::vector<double> A(tiles * 64), B(tiles * 64), C(tiles * 256), D(tiles * 256);
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  std::uniform_int_distribution<int> e(-30, 30);
  for (int t = 0; t < tiles; ++t) {
    const int mode = t % 4;  // 0: O(1) values; 1: wide exponent spread; 2: cancelling pairs; 3: zeros in some operands (the sparsity of J_u)
    for (int i = 0; i < 64; ++i) {
      A[t * 64 + i] = u(g) * (mode == 1 ? std::ldexp(1.0, e(g)) : 1.0);
      B[t * 64 + i] = u(g) * (mode == 1 ? std::ldexp(1.0, e(g)) : 1.0);
    }
    for (int i = 0; i < 256; ++i) C[t * 256 + i] = u(g) * (mode == 1 ? std::ldexp(1.0, e(g)) : 1.0);
    if (mode == 2)
      for (int j = 0; j < 16; ++j) {  // a1 b1 ~ -a0 b0 up to the last bits, c tiny: the order of the additions decides the result
        for (int jj = 0; jj < 16; ++jj) C[t * 256 + j * 16 + jj] *= 1e-14;
        A[t * 64 + j * 4 + 1] = -A[t * 64 + j * 4 + 0] * (1.0 + 3e-16 * (j + 1));
      }
    if (mode == 2)
      for (int j = 0; j < 16; ++j) B[t * 64 + 1 * 16 + j] = B[t * 64 + 0 * 16 + j];
    if (mode == 3)
      for (int j = 0; j < 16; ++j) {
        A[t * 64 + j * 4 + (j & 3)] = 0.0;
        if (j & 1) B[t * 64 + (j & 3) * 16 + j] = 0.0;
      }
  }
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
  k<<<tiles, 64>>>(dA, dB, dC, dD, tiles);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
  long n = 0, bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_exact = 0, seq_ne_rev = 0, seq_ne_pair = 0;
  long bad_by_mode[4] = {0, 0, 0, 0};
  for (int t = 0; t < tiles; ++t)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double *a = &A[t * 64 + i * 4];
        double b[4];
        for (int q = 0; q < 4; ++q) b[q] = B[t * 64 + q * 16 + j];
        const double c = C[t * 256 + i * 16 + j], d = D[t * 256 + i * 16 + j];
        const double seq = std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], std::fma(a[0], b[0], c))));
        const double rev = std::fma(a[0], b[0], std::fma(a[1], b[1], std::fma(a[2], b[2], std::fma(a[3], b[3], c))));
        const double pair = c + (std::fma(a[0], b[0], a[1] * b[1]) + std::fma(a[2], b[2], a[3] * b[3]));
        const long double ex = (long double)c + (long double)a[0] * b[0] + (long double)a[1] * b[1] + (long double)a[2] * b[2] + (long double)a[3] * b[3];
        ++n;
        if (std::memcmp(&d, &seq, 8)) { ++bad_seq; ++bad_by_mode[t % 4]; }
        if (std::memcmp(&d, &rev, 8)) ++bad_rev;
        if (std::memcmp(&d, &pair, 8)) ++bad_pair;
        const double exd = (double)ex;
        if (std::memcmp(&d, &exd, 8)) ++bad_exact;
        if (std::memcmp(&seq, &rev, 8)) ++seq_ne_rev;
        if (std::memcmp(&seq, &pair, 8)) ++seq_ne_pair;
      }
  printf("v_mfma_f64_16x16x4_f64 against candidate arithmetics, %ld elements (%d tiles: O(1) values / wide exponents / cancelling / sparse)\n", n, tiles);
  printf("  differs from seq   (fma chain k = 0..3 on the accumulator): %ld   (by operand family: %ld %ld %ld %ld)\n", bad_seq, bad_by_mode[0],
         bad_by_mode[1], bad_by_mode[2], bad_by_mode[3]);
  printf("  differs from rev   (fma chain k = 3..0):                    %ld\n", bad_rev);
  printf("  differs from pair  (c + ((p0 + p1) + (p2 + p3))):           %ld\n", bad_pair);
  printf("  differs from exact (long double sum, rounded once):         %ld\n", bad_exact);
  printf("  (the candidates part from each other: seq != rev in %ld, seq != pair in %ld elements)\n", seq_ne_rev, seq_ne_pair);
  return bad_seq ? 2 : 0;
}
