// backward_common.h -- the per-knot pieces every backward kernel shares (ilqr.hh:97-147): lane selects, butterflies, the row gather and DPP
// broadcasts of Q_uu, the two tile products T = V M and H = C + M^T T on the fp64 matrix core, the unpivoted LDL^T and its substitution.
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "kernels_common.h"

namespace qilqr {

// ---------------------------------------------------------------------------------------------
// k_backward: one wavefront per trajectory (block = 64 threads).  See backward_layout.h.
// force = 1: run on every trajectory, no convergence test (the stand-alone backwards_pass API).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sel4(const double v[4], int kk) {
  const double lo = (kk & 1) ? v[1] : v[0], hi = (kk & 1) ? v[3] : v[2];
  return (kk & 2) ? hi : lo;
}
// 1/x for the pivots of the unpivoted LDL^T: the hardware estimate (4.6e-8 relative) and two Newton steps (1.1e-16;
// profiles/microbench/rcp_accuracy.hip): half the dependent depth of the IEEE division sequence.  ONE step (2.2e-15) was measured in round 5
// (-DQILQR_RCP_NEWTON_STEPS=1): the second step is two dependent fp64 instructions behind each of a knot's four pivots, and without it a
// launch takes 65.0 instead of 66.8 us per 100 knots, a configs[1] solve 4.66 instead of 4.75 ms (+ 1.8 %), with every solver-against-oracle
// maximum of WHOLE solves unchanged (cost 1e-13, trajectory 1.7e-9 at 200 knots) -- but ONE backward pass at 200 knots lands 3.9e-9 of the
// largest gain from the oracle's instead of 1e-9 (what a knot injects is carried to the pass's end almost undamped), past the bar
// tests/test_gpu_parity.py holds that pass to.  Parity before 1.8 %: two steps stay.
#ifndef QILQR_RCP_NEWTON_STEPS
#define QILQR_RCP_NEWTON_STEPS 2
#endif
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#if QILQR_RCP_NEWTON_STEPS >= 2
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#endif
  return r;
}
// value of x in lane `src` (compile-time constant), broadcast to the wave
__device__ __forceinline__ double bcast_lane(double x, int src) {
  const long long v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)v, src);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// x + (x of the lane 16 / 32 positions away): the two butterfly steps of a sum over the four 16-lane
// rows, with v_permlane16_swap / v_permlane32_swap (VALU, no LDS round trip)
__device__ __forceinline__ double xor16_sum(double x) {
  const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
}
__device__ __forceinline__ double xor32_sum(double x) {
  const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
}

// r[a] = value of x in the lane of the same column and row a (a = 0..3), for every lane: three
// permlane swaps per dword instead of four ds_bpermute round trips
__device__ __forceinline__ void gather_rows(double x, double r[4]) {
  const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // {x0,x0,x2,x2}, {x1,x1,x3,x3}
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const auto la = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);  // x0 everywhere, x2 everywhere
  const auto ha = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
  const auto lb = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);  // x1, x3
  const auto hb = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
  r[0] = __longlong_as_double(((long long)ha[0] << 32) | la[0]);
  r[1] = __longlong_as_double(((long long)hb[0] << 32) | lb[0]);
  r[2] = __longlong_as_double(((long long)ha[1] << 32) | la[1]);
  r[3] = __longlong_as_double(((long long)hb[1] << 32) | lb[1]);
}

// value of x in lane SRC of the caller's own row of 16 lanes (DPP row_newbcast: one v_mov_b64_dpp, no trip
// through the scalar registers)
template <int SRC>
__device__ __forceinline__ double row_bcast(double x) {
  return __builtin_amdgcn_mov_dpp(x, 0x150 + SRC, 0xf, 0xf, false);  // no `old` operand: nothing to zero or copy first
}
template <int A>
__device__ __forceinline__ void bcast_quu_row(const double col[4], double ghat, double Quu[16], double Qu[4]) {
  // row A of the lower triangle of Q_uu and Q_u[A]; the four rows of 16 lanes hold identical copies of
  // col[] and ghat in their lanes 12..15, so a broadcast inside each row reaches the whole wave
  Quu[A * 4 + 0] = row_bcast<12>(col[A]);
  if constexpr (A >= 1) Quu[A * 4 + 1] = row_bcast<13>(col[A]);
  if constexpr (A >= 2) Quu[A * 4 + 2] = row_bcast<14>(col[A]);
  if constexpr (A >= 3) Quu[A * 4 + 3] = row_bcast<15>(col[A]);
  Qu[A] = row_bcast<12 + A>(ghat);
}

// ---- the per-knot pieces every backward kernel shares (stated once; each kernel inlines them) ----------------------------
// T = V M: three fp64 MFMAs over the contraction index 4 kc + kk (A = V_xx in A layout, B = M = [J_x | J_u])
__device__ __forceinline__ d4 bw_tile_T(const double (&va)[3], const double (&m)[3]) {
  d4 T = {0.0, 0.0, 0.0, 0.0};
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[0], m[0], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[1], m[1], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[2], m[2], T, 0, 0, 0);
  return T;
}
// H = blkdiag(C_xx, C_uu) + M^T T  (ilqr.hh:118-124 in one accumulator tile): M^T in A layout is the same three registers
// as M in B layout, and T's result registers are the B operand
__device__ __forceinline__ d4 bw_tile_H(const double (&m)[3], const d4 &T, const double (&cx)[3], double cuu) {
  d4 H = {cx[0], cx[1], cx[2], cuu};
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[0], T[0], H, 0, 0, 0);
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[1], T[1], H, 0, 0, 0);
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[2], T[2], H, 0, 0, 0);
  return H;
}
// the same with the products in the order kc = 2, 0, 1 -- the order of bw4_fused_wave (rows 12..15 of H receive nothing from the other two
// products, so that register is final one product early there) and, since round 6, of every k_backward4 form: the order is part of the
// arithmetic of rows 0..11
__device__ __forceinline__ d4 bw_tile_H201(const double (&m)[3], const d4 &T, const double (&cx)[3], double cuu) {
  d4 H = {cx[0], cx[1], cx[2], cuu};
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[2], T[2], H, 0, 0, 0);
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[0], T[0], H, 0, 0, 0);
  H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[1], T[1], H, 0, 0, 0);
  return H;
}
// (Round 3 tried to take the factorisation off the matrix instructions' chain: rows 12..15 of H -- [Q_ux | Q_uu] -- are complete
// after the kc = 2 product alone, because J_u is zero in rows 0..7, so the gather, the LDL^T and the solve could run while the
// other two products execute.  It does not pay and cannot: v_mfma_f64 and the fp64 vector instructions use the SAME double-
// precision units of a SIMD (profiles/microbench/coissue.hip: an fp64 FMA beside an fp64 MFMA takes 8.7 cycles instead of 4.7),
// so interleaving them gains nothing -- 72.0 us per launch against 65.4 with one live trajectory per block -- and the
// compiler's schedule for it also reused the matrix instruction's source tile for vector results while it was in flight
// (results NaN).  A matrix wave's floor is its fp64 work: 7 x 64 cycles of MFMA plus ~60 fp64 vector instructions, plus the
// latencies between them; the 23 integer / move instructions the unrolled loop removes ride in their shadow.)
// LDL^T of the lower triangle of Q_uu WITHOUT pivoting (the symmetric-weight kernels: Q_uu = 2 R + J_u^T V_xx J_u is positive
// definite there; Eigen's LDLT, ilqr.hh:126, pivots on the diagonal -- the same factors in exact arithmetic; the general kernel
// pivots, backward_layout.h).  Reciprocals of the pivots by rcp_nr.
struct Ldlt4 {
  double l10, l20, l30, l21, l31, l32, i0, i1, i2, i3;
};
__device__ __forceinline__ Ldlt4 ldlt4_factor(const double (&Quu)[16]) {
  Ldlt4 f;
  f.i0 = rcp_nr(Quu[0]);
  f.l10 = Quu[4] * f.i0; f.l20 = Quu[8] * f.i0; f.l30 = Quu[12] * f.i0;
  const double d1 = Quu[5] - f.l10 * Quu[4];
  f.i1 = rcp_nr(d1);
  const double c21 = Quu[9] - f.l20 * Quu[4], c31 = Quu[13] - f.l30 * Quu[4];
  f.l21 = c21 * f.i1; f.l31 = c31 * f.i1;
  const double d2 = Quu[10] - f.l20 * Quu[8] - f.l21 * c21;
  f.i2 = rcp_nr(d2);
  const double c32 = Quu[14] - f.l30 * Quu[8] - f.l31 * c21;
  f.l32 = c32 * f.i2;
  const double d3 = Quu[15] - f.l30 * Quu[12] - f.l31 * c31 - f.l32 * c32;
  f.i3 = rcp_nr(d3);
  return f;
}
// x = -Q_uu^-1 rhs with those factors: a column of K (ilqr.hh:127) or the feed-forward k (:128)
// (Solved for the right-hand side -r: the signs ride on the operands of the multiply-adds instead of four negations at the end.
// fma(-a, b, -c) = -fma(a, b, c) exactly, so every intermediate is the exact negative of the plain solve's and the result has
// the same bits.)
__device__ __forceinline__ void ldlt4_solve_neg(const Ldlt4 &f, double r0, double r1, double r2, double r3, double (&x)[4]) {
  const double y1 = __builtin_fma(f.l10, r0, -r1);                                                        // y0 = -r0
  const double y2 = __builtin_fma(-f.l21, y1, __builtin_fma(f.l20, r0, -r2));
  const double y3 = __builtin_fma(-f.l32, y2, __builtin_fma(-f.l31, y1, __builtin_fma(f.l30, r0, -r3)));
  const double x3 = y3 * f.i3;
  const double x2 = __builtin_fma(-f.l32, x3, y2 * f.i2);
  const double x1 = __builtin_fma(-f.l31, x3, __builtin_fma(-f.l21, x2, y1 * f.i1));
  const double x0 = __builtin_fma(-f.l30, x3, __builtin_fma(-f.l20, x2, __builtin_fma(-f.l10, x1, -(r0 * f.i0))));
  x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3;
}

// one slot of the LDS operand rings of k_backward2 / k_backward4 / k_solve4
constexpr int BW2_REC = 128;                   // doubles reserved for a record (symmetric layouts: stride <= 128)
constexpr int BW2_BUF = BW2_REC + CTAB_SIZE;   // one ring slot: record, then the constant operand table

}  // namespace qilqr
