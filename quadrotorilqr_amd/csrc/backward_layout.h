// backward_layout.h -- where each lane of the backward-pass wavefront finds its operands.
//
// The backward kernels (k_backward, and the matrix wavefronts of k_backward2 / k_backward4) give one 64-lane
// wavefront to one trajectory's matrix recursion and contract the per-knot matrices on the fp64 matrix core: v_mfma_f64_16x16x4_f64, D(16x16) = A(16x4) B(4x16) + C.
// Lane l = (j = l & 15, kk = l >> 4) supplies A[j][kk] and B[kk][j]; it receives
// D[4 r + kk][j] in result register r (r = 0..3).  With the stacked Jacobian
//        M = [ J_x | J_u ]          (12 x 16: the control Jacobian fills the tile exactly)
// one knot of the Riccati recursion is
//        T = V M          3 MFMA  (A = V, B = M, contraction index 4 kc + kk, kc = 0..2)
//        H = C + M^T T    3 MFMA  (A = M^T, B = T: T's result registers ARE the B operand)
// and H = [[Q_xx, Q_xu],[Q_ux, Q_uu]].  M^T in A layout is the same three registers as M in
// B layout, so a lane holds exactly three elements of M: rows kk, 4+kk, 8+kk of column j.
//
// m_source() says where such an element lives in the knot record written by k_linearize
// (se3_math.h, LIN_*), or that it is a constant (zero, identity or the constant J_u).
#pragma once
#include "se3_math.h"

namespace qilqr {

// element (row, col) of M = [J_x | J_u], row < 12, col < 16.
// Returns the offset into the knot record, or -1 with *cst = the constant value.
QILQR_HD int m_source(int row, int col, const double *Bu, double *cst) {
  *cst = 0.0;
  if (col >= 12) {
    *cst = Bu[row * 4 + (col - 12)];
    return -1;
  }
  const int br = row / 3, rr = row % 3, bc = col / 3, cc = col % 3;
  int blk = -1;
  if (br == 0) {
    blk = bc;  // E^T | Ad top-right | dt Jr | dt Q
  } else if (br == 1) {
    if (bc == 1) blk = 0;       // E^T
    else if (bc == 3) blk = 2;  // dt Jr
  } else if (br == 2) {
    if (bc == 1) blk = 4;  // -dt g hat(R^T e_z)
    else if (bc == 2) {
      *cst = (rr == cc) ? 1.0 : 0.0;
      return -1;
    }
  } else {
    if (bc == 3) blk = 5;  // I + dt D
  }
  if (blk < 0) return -1;
  return LIN_BLK + blk * 9 + rr * 3 + cc;
}

// Constant operand table (device memory, built once per solver): entries that do not change from
// knot to knot are read through the same unconditional loads as the record entries, so that the
// loads of knot i-1 can be issued before the chain of knot i and nothing selects on their result.
//   [0] = 0, [1] = 1, [2..49] = J_u (12x4), [50..85] = 2 Q[6:12, 6:12] (the velocity block of C_xx, the only constant part
//   of it that a symmetric layout reads from the table: cxx_source_tab).  86 entries: the kernels that stage records
//   through LDS keep a copy behind every ring slot, and at 194 entries (all of 2 Q, round 1) k_backward4's 46 KB of LDS
//   allowed three blocks per CU where 33 KB allow four
constexpr int CTAB_ZERO = 0, CTAB_ONE = 1, CTAB_BU = 2, CTAB_2Q = 50, CTAB_SIZE = 86;
QILQR_HD void build_ctab(const double *Bu, const double *Q, double *tab) {
  tab[CTAB_ZERO] = 0.0;
  tab[CTAB_ONE] = 1.0;
  for (int i = 0; i < 48; ++i) tab[CTAB_BU + i] = Bu[i];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) tab[CTAB_2Q + r * 6 + c] = 2.0 * Q[(6 + r) * 12 + 6 + c];
}
// the same for a record of layout L: with the dense M of the Runge-Kutta extension (RecLayout.dense_m) every element of
// M is an entry of the record -- J_u too, it depends on the state there
QILQR_HD int m_source(const RecLayout &L, int row, int col, const double *Bu, double *cst) {
  if (L.dense_m) {
    *cst = 0.0;
    return LIN_BLK + row * 16 + col;
  }
  return m_source(row, col, Bu, cst);
}
// like m_source, but constants are returned as an index into the table: -1 - index
QILQR_HD int m_source_tab(int row, int col) {
  if (col >= 12) return -1 - (CTAB_BU + row * 4 + (col - 12));
  double cst;
  const int off = m_source(row, col, nullptr, &cst);
  if (off >= 0) return off;
  return -1 - (cst == 1.0 ? CTAB_ONE : CTAB_ZERO);
}
QILQR_HD int m_source_tab(const RecLayout &L, int row, int col) {
  return L.dense_m ? LIN_BLK + row * 16 + col : m_source_tab(row, col);
}
QILQR_HD int cxx_source_tab(const RecLayout &L, int row, int col) {
  if (!L.sym) return L.off_cxx + row * 12 + col;
  const int i = row < col ? row : col, j = row < col ? col : row;
  if (j < 6) return L.off_cxx + symrow_index(!L.ur_zero, i, j);
  if (i < 6) return L.ur_zero ? -1 - CTAB_ZERO : L.off_cxx + symrow_index(true, i, j);
  return -1 - (CTAB_2Q + (row - 6) * 6 + (col - 6));  // (i >= 6 here: both indices are in the velocity block)
}

// Eigen 3.4.0 LDLT<Matrix4d, Lower> as the reference uses it (ilqr.hh:126-128: Q_uu.ldlt().solve(rhs)): in place on the
// lower triangle, DIAGONAL PIVOTING (the largest |d_ii| of the trailing block, the first one on ties), x = P^T L^-T D^-1
// L^-1 P b.  Every index is a compile-time constant (the sequence of transpositions is dispatched over its 24 possible
// values), so the 4x4 stays in registers.  Used by the general kernel (k_backward<false>: non-symmetric weights,
// or force_general = 1), where faithfulness to the reference is the point; the symmetric-weight kernels factor without
// pivoting (identical in exact arithmetic when Q_uu is positive definite).
// a / b for the quotients of the factorisation.  The sequence the compiler emits for an IEEE `/` on gfx950 is ~ 40 dependent fp64
// instructions (v_div_scale, v_rcp, four refinements, v_div_fmas, v_div_fixup), and a knot has ten of them: half of the general kernel's
// instructions.  On the device the divisions by one pivot share ONE reciprocal (hardware estimate + one Newton step: 2.2e-15 relative)
// and every quotient gets one correction step, q = a r, q <- q + r (a - b q) with the residual exact in a fused multiply-add -- a Newton
// step for the quotient itself, which squares the reciprocal's error: a FAITHFULLY rounded quotient (the IEEE quotient or its neighbour: with
// a reciprocal good to ~2^-48 a misrounding has a probability of the order of 2^-43 per quotient) that EQUALLED the IEEE quotient in every
// one of 2 x 10^7 random cases over five magnitudes (profiles/microbench/rcp_accuracy.hip) -- not Eigen's division by proof, Eigen's division
// in every sampled case.  Outside |b| in about [1e-290, 1e290] the reciprocal or a r leaves the representable range where a / b is
// representable (Eigen gives finite values there, this gives inf or NaN); Eigen's own threshold for a zero pivot is 2.2e-308, and a Q_uu with
// pivots of 1e-290 has no usable gains in the reference either.  The host (tests/host_harness.cpp, where
// the oracle's pivoted LDL^T is compared bit for bit) divides.
struct PivotRcp {
  double b, r;
};
QILQR_HD PivotRcp pivot_rcp(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);                   // 4.6e-8 relative (profiles/microbench/rcp_accuracy.hip)
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);   // 2.2e-15: enough, the quotient's own correction step squares it
  return PivotRcp{b, r};
#else
  return PivotRcp{b, 0.0};
#endif
}
QILQR_HD double div_by(double a, const PivotRcp &p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double q = a * p.r;
  return __builtin_fma(__builtin_fma(-p.b, q, a), p.r, q);
#else
  return a / p.b;
#endif
}
// Eigen's pivot search looks at mat.diagonal().tail(size - k) BEFORE column k is updated, and its left-looking update touches only column
// k: the diagonal entries it compares are the ORIGINAL ones (exchanged along with the rows).  The transpositions are therefore known
// from the four |d_ii| alone -- a selection sort with Eigen's tie rule (the first of equal magnitudes, in the arrangement the earlier
// exchanges left) -- before any arithmetic: b0 in 0..3, b1 in 1..3, b2 in 2..3.
QILQR_HD void ldlt4_pivot_order(const double (&A)[16], int &b0, int &b1, int &b2) {
  double d0 = fabs(A[0]), d1 = fabs(A[5]), d2 = fabs(A[10]), d3 = fabs(A[15]);
  b0 = 0;
  double best = d0;
  if (d1 > best) { best = d1; b0 = 1; }
  if (d2 > best) { best = d2; b0 = 2; }
  if (d3 > best) { best = d3; b0 = 3; }
  const double o0 = d0;  // rows / columns 0 and b0 change places: so do their diagonal entries
  d1 = (b0 == 1) ? o0 : d1; d2 = (b0 == 2) ? o0 : d2; d3 = (b0 == 3) ? o0 : d3;
  b1 = 1;
  best = d1;
  if (d2 > best) { best = d2; b1 = 2; }
  if (d3 > best) { best = d3; b1 = 3; }
  const double o1 = d1;
  d2 = (b1 == 2) ? o1 : d2; d3 = (b1 == 3) ? o1 : d3;
  b2 = (d3 > d2) ? 3 : 2;
}
// position i of the permuted matrix holds row / column ldlt4_perm_at(b0, b1, b2, i) of the original: the transpositions (0 b0)(1 b1)(2 b2)
// applied in that order to the identity
constexpr int ldlt4_perm_at(int b0, int b1, int b2, int i) {
  int p[4] = {0, 1, 2, 3};
  int t = p[0]; p[0] = p[b0]; p[b0] = t;
  t = p[1]; p[1] = p[b1]; p[b1] = t;
  t = p[2]; p[2] = p[b2]; p[b2] = t;
  return p[i];
}
// The factorisation and the solves for ONE sequence of transpositions, known at compile time: the exchanges of rows, columns and
// right-hand side entries are a renaming of registers (until round 5 they were ~ 100 moves behind data-dependent branches per knot, and the
// right-hand side lived in scratch memory).  Element (i, j), i >= j, of the exchanged lower triangle is element (max, min) of the original
// one at the permuted indices -- what Eigen's exchanges restricted to the lower triangle produce.  The arithmetic is Eigen's, operation by
// operation: for every column k the products with the earlier columns (temp = D(0..k) A10^T; a_kk -= A10 temp; A21 -= A20 temp), then
// A21 /= a_kk if a_kk != 0; L y = P b; y_i /= d_i (0 where |d_i| <= numeric_limits::min()); L^T z = y; x = P^T z.
template <int B0, int B1, int B2>
QILQR_HD void ldlt4_solve_permuted(const double (&A)[16], const double (&rhs)[4], double (&x)[4]) {
  constexpr int P[4] = {ldlt4_perm_at(B0, B1, B2, 0), ldlt4_perm_at(B0, B1, B2, 1), ldlt4_perm_at(B0, B1, B2, 2), ldlt4_perm_at(B0, B1, B2, 3)};
  double m[16], y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) m[i * 4 + j] = A[(P[i] > P[j] ? P[i] : P[j]) * 4 + (P[i] > P[j] ? P[j] : P[i])];
    y[i] = rhs[P[i]];
  }
#pragma unroll
  for (int K = 0; K < 4; ++K) {
    if (K > 0) {
      double temp[3];
#pragma unroll
      for (int jj = 0; jj < K; ++jj) temp[jj] = m[jj * 4 + jj] * m[K * 4 + jj];
      double sacc = 0.0;
#pragma unroll
      for (int jj = 0; jj < K; ++jj) sacc += m[K * 4 + jj] * temp[jj];
      m[K * 4 + K] -= sacc;
#pragma unroll
      for (int ii = K + 1; ii < 4; ++ii) {
        double r = 0.0;
#pragma unroll
        for (int jj = 0; jj < K; ++jj) r += m[ii * 4 + jj] * temp[jj];
        m[ii * 4 + K] -= r;
      }
    }
    const double akk = m[K * 4 + K];
    const bool valid = fabs(akk) > 0.0;  // (Eigen: `if (rs > 0 && pivot_is_valid) A21 /= realAkk`; a selection here, not a branch)
    const PivotRcp pr = pivot_rcp(akk);
#pragma unroll
    for (int ii = K + 1; ii < 4; ++ii) m[ii * 4 + K] = valid ? div_by(m[ii * 4 + K], pr) : m[ii * 4 + K];
  }
#pragma unroll
  for (int ii = 0; ii < 4; ++ii)
#pragma unroll
    for (int jj = 0; jj < ii; ++jj) y[ii] -= m[ii * 4 + jj] * y[jj];
  const double tol = 2.2250738585072014e-308;  // numeric_limits<double>::min(), Eigen's threshold for a zero pivot
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) y[ii] = (fabs(m[ii * 4 + ii]) > tol) ? div_by(y[ii], pivot_rcp(m[ii * 4 + ii])) : 0.0;
#pragma unroll
  for (int ii = 3; ii >= 0; --ii)
#pragma unroll
    for (int jj = ii + 1; jj < 4; ++jj) y[ii] -= m[jj * 4 + ii] * y[jj];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[P[i]] = y[i];
#if defined(__HIP_DEVICE_COMPILE__)
  // (the four results in vector registers before the 24 instances meet: their stores x[P[i]] = y[i] would otherwise be merged into one
  // store through a run-time index, and x would live in scratch memory)
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#endif
}
template <int B0, int B1>
QILQR_HD void ldlt4_dispatch2(int b2, const double (&A)[16], const double (&rhs)[4], double (&x)[4]) {
  if (b2 == 2) ldlt4_solve_permuted<B0, B1, 2>(A, rhs, x);
  else ldlt4_solve_permuted<B0, B1, 3>(A, rhs, x);
}
template <int B0>
QILQR_HD void ldlt4_dispatch1(int b1, int b2, const double (&A)[16], const double (&rhs)[4], double (&x)[4]) {
  if (b1 == 1) ldlt4_dispatch2<B0, 1>(b2, A, rhs, x);
  else if (b1 == 2) ldlt4_dispatch2<B0, 2>(b2, A, rhs, x);
  else ldlt4_dispatch2<B0, 3>(b2, A, rhs, x);
}
// x = Q_uu^-1 rhs as Eigen's LDLT<Matrix4d, Lower> computes it.  Q_uu must be the same in every lane of the wavefront (the right-hand
// sides may differ): the pivot order is then wave-uniform and the dispatch over its 24 values a few scalar branches.
QILQR_HD void ldlt4_pivoted_solve(const double (&Quu)[16], const double (&rhs)[4], double (&x)[4]) {
  int b0, b1, b2;
  ldlt4_pivot_order(Quu, b0, b1, b2);
#if defined(__HIP_DEVICE_COMPILE__)
  const int code = __builtin_amdgcn_readfirstlane(b0 | (b1 << 2) | (b2 << 4));
  b0 = code & 3; b1 = (code >> 2) & 3; b2 = code >> 4;
#endif
  if (b0 == 0) ldlt4_dispatch1<0>(b1, b2, Quu, rhs, x);
  else if (b0 == 1) ldlt4_dispatch1<1>(b1, b2, Quu, rhs, x);
  else if (b0 == 2) ldlt4_dispatch1<2>(b1, b2, Quu, rhs, x);
  else ldlt4_dispatch1<3>(b1, b2, Quu, rhs, x);
}

}  // namespace qilqr
