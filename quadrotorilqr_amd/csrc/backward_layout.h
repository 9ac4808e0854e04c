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

}  // namespace qilqr
