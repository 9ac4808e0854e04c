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
// L^-1 P b with IEEE divisions.  Every index is a compile-time constant (the pivot position is dispatched over its three
// possible values), so the 4x4 stays in registers.  Used by the general kernel (k_backward<false>: non-symmetric weights,
// or force_general = 1), where faithfulness to the reference is the point; the symmetric-weight kernels factor without
// pivoting (identical in exact arithmetic when Q_uu is positive definite).
// a / b for the quotients of the factorisation.  The sequence the compiler emits for an IEEE `/` on gfx950 is ~ 40 dependent fp64
// instructions (v_div_scale, v_rcp, four refinements, v_div_fmas, v_div_fixup), and a knot has ten of them: half of the general kernel's
// instructions.  On the device the divisions by one pivot share ONE reciprocal (hardware estimate + two Newton steps: within an ulp of
// 1 / b) and every quotient gets one correction step, q = a r, q <- q + r (a - b q) with the residual exact in a fused multiply-add: the
// correctly rounded quotient except for rare ties of the last correction (then the neighbouring double), i.e. Eigen's division to the
// last bit or the one beside it.  Pivots of magnitude below ~ 1e-292 (the reciprocal overflows) are beyond it; Eigen's own threshold
// for a zero pivot is 2.2e-308, and such a Q_uu has no usable gains in the reference either.  The host (tests/host_harness.cpp, where
// the oracle's pivoted LDL^T is compared bit for bit) divides.
struct PivotRcp {
  double b, r;
};
QILQR_HD PivotRcp pivot_rcp(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
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
// y[k] <-> y[big] (big in k..3, wave-uniform).  Three `if (big == B) swap(y[k], y[B])` in a row -- and every way of writing the same
// with selections -- are merged by the compiler into ONE access y[big] with a run-time index, which keeps the whole array in scratch
// memory: the right-hand side's six trips there sat on the knot's dependent chain until round 5.  On the device the three exchanges are
// therefore kept apart by empty asm statements that claim the four values in vector registers.
QILQR_HD void ldlt4_swap_rhs(double (&y)[4], int k, int big) {
#if defined(__HIP_DEVICE_COMPILE__)
#define QILQR_PIN_Y() asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]))
#else
#define QILQR_PIN_Y() do { } while (0)
#endif
  if (big != k) {
    QILQR_PIN_Y();
    if (big == 1) { const double t = y[k]; y[k] = y[1]; y[1] = t; }
    QILQR_PIN_Y();
    if (big == 2) { const double t = y[k]; y[k] = y[2]; y[2] = t; }
    QILQR_PIN_Y();
    if (big == 3) { const double t = y[k]; y[k] = y[3]; y[3] = t; }
    QILQR_PIN_Y();
  }
#undef QILQR_PIN_Y
}
template <int K, int BIG>
QILQR_HD void ldlt4_swap(double (&m)[16]) {
  // symmetric exchange of rows / columns K and BIG restricted to the lower triangle (the right-hand side: ldlt4_swap_rhs)
#pragma unroll
  for (int jj = 0; jj < K; ++jj) { const double t = m[K * 4 + jj]; m[K * 4 + jj] = m[BIG * 4 + jj]; m[BIG * 4 + jj] = t; }
#pragma unroll
  for (int ii = BIG + 1; ii < 4; ++ii) { const double t = m[ii * 4 + K]; m[ii * 4 + K] = m[ii * 4 + BIG]; m[ii * 4 + BIG] = t; }
  { const double t = m[K * 4 + K]; m[K * 4 + K] = m[BIG * 4 + BIG]; m[BIG * 4 + BIG] = t; }
#pragma unroll
  for (int ii = K + 1; ii < BIG; ++ii) { const double t = m[ii * 4 + K]; m[ii * 4 + K] = m[BIG * 4 + ii]; m[BIG * 4 + ii] = t; }
}
template <int K>
QILQR_HD int ldlt4_pivot_step(double (&m)[16], double (&y)[4]) {
  int bigv = K;
  double best = fabs(m[K * 4 + K]);
#pragma unroll
  for (int ii = K + 1; ii < 4; ++ii)
    if (fabs(m[ii * 4 + ii]) > best) { best = fabs(m[ii * 4 + ii]); bigv = ii; }
  int big = bigv;
#if defined(__HIP_DEVICE_COMPILE__)
  // every lane factors the same Q_uu: the pivot is wave-uniform, and with it in a scalar register the exchange of rows and columns is
  // behind scalar branches
  big = __builtin_amdgcn_readfirstlane(bigv);
#endif
  if constexpr (K < 1) { if (big == 1) ldlt4_swap<K, 1>(m); }
  if constexpr (K < 2) { if (big == 2) ldlt4_swap<K, 2>(m); }
  if constexpr (K < 3) { if (big == 3) ldlt4_swap<K, 3>(m); }
  if constexpr (K < 3) ldlt4_swap_rhs(y, K, big);
  if constexpr (K > 0) {
    double temp[K > 0 ? K : 1];
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
  if (fabs(akk) > 0.0) {
    const PivotRcp pr = pivot_rcp(akk);
#pragma unroll
    for (int ii = K + 1; ii < 4; ++ii) m[ii * 4 + K] = div_by(m[ii * 4 + K], pr);
  }
  return big;
}
// x = Q_uu^-1 rhs.  (The permutation is applied to the right-hand side as the factorisation finds it: the same
// transpositions, in the same order, that Eigen applies to b before the triangular solves.)
QILQR_HD void ldlt4_pivoted_solve(const double (&Quu)[16], const double (&rhs)[4], double (&x)[4]) {
  double m[16], y[4];
#pragma unroll
  for (int e = 0; e < 16; ++e) m[e] = Quu[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = rhs[e];
  const int t0 = ldlt4_pivot_step<0>(m, y);
  const int t1 = ldlt4_pivot_step<1>(m, y);
  const int t2 = ldlt4_pivot_step<2>(m, y);
  (void)ldlt4_pivot_step<3>(m, y);
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
  // P^T: the transpositions in reverse order
  ldlt4_swap_rhs(y, 2, t2);
  ldlt4_swap_rhs(y, 1, t1);
  ldlt4_swap_rhs(y, 0, t0);
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = y[e];
}

}  // namespace qilqr
