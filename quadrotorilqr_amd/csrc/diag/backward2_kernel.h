// backward2_kernel.h -- k_backward2: a matrix and a gradient wavefront per trajectory (diagnostics build only: force_general = 3).
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "../backward_common.h"

namespace qilqr {

// ---------------------------------------------------------------------------------------------
// k_backward2: the recursion for symmetric weights with TWO cooperating wavefronts per trajectory
// (block = 128).  The value gradient V_x never feeds back into V_xx, so its part of every knot is
// taken off the serial chain of the matrix recursion:
//   wave M (matrix):   T = V M, H = C + M^T T, LDL^T of Q_uu, K = -Quu^-1 Q_ux, V_xx = Q_xx + Q_xu K;
//                      stores K; hands K and the LDL^T factors to G through LDS
//   wave G (gradient): one knot behind.  [Q_x ; Q_u] = [C_x ; C_u] + M^T V_x, k = -Quu^-1 Q_u with
//                      M's factors, V_x = Q_x + K^T Q_u, Q_u^T k; stores k.  It also streams the knot
//                      records from HBM into a three-deep LDS ring (two coalesced loads per knot), from
//                      which both waves take their operands (M one knot ahead, into registers).
// Interval I_i (between two barriers) for i = n-1 .. 0:
//   M: knot i (operands in registers); reads knot i-1's operands from ring[(i-1) % 3]; writes K_i, factors_i
//   G: gradient step of knot i+1 (ring[(i+1) % 3], kf[(i+1) & 1]); then record i-2 -> ring[(i-2) % 3];
//      then issues the loads of record i-3
// Same arithmetic as k_backward<true>: the gains are bit-identical.
// ---------------------------------------------------------------------------------------------
#ifdef QILQR_WITH_BACKWARD2  // diagnostics build only (make diag): the product takes k_backward4 at every batch size up to 8192
template <typename S>
__global__ __launch_bounds__(128) void k_backward2(ModelConsts<double> c, SolveParams p, BatchState st, int B, int n,
                                                   int force) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0: M, 1: G
  __shared__ double cost_scr[2][64];  // the settle step's knot costs, one row per wave
  // ---- settle the pending candidate (ilqr.hh:70-84, 174-194).  Both waves take the decision from the
  // same global data; wave G's lane 0 applies it after a barrier (nobody reads those words afterwards).
  int fl = st.flags[b];
  int cur = st.cur[b];
  const int it0 = st.iters[b];
  const int trial0 = st.trial[b];
  const double prev_cost0 = st.prev_cost[b], alpha0 = st.alpha[b];
  const double term0 = st.terms[2 * b], term1 = st.terms[2 * b + 1];
  double cost_now = st.cost[b];
  double mu = (p.mu_init > 0.0) ? st.mu[b] : 0.0;
  bool restart = false;
  bool settle = false, accept = false, count_active = false;
  int status = -1;
  double new_cost = 0.0;
  if (!force) {
    if (fl & F_SEARCH) {
      settle = true;
      const double *kc = st.knot_cost[cur ^ 1];
      double *scr = cost_scr[role];  // through LDS, every lane adding in order from broadcast reads (see k_backward4)
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const int cnt = (n - base < 64) ? n - base : 64;
        scr[lane] = (i < n) ? kc[cost_index(b, i, n)] : 0.0;
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
          double x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = scr[t + e];
#pragma unroll
          for (int e = 0; e < 8; ++e) new_cost += x[e];
        }
        for (; t < cnt; ++t) new_cost += scr[t];
      }
      if (it0 == 0) {
        accept = true;  // ilqr.hh:71-73
      } else {
        const double desired = p.reduction_frac * cost_reduction(term0, term1, alpha0);
        accept = (new_cost - prev_cost0 < desired);  // ilqr.hh:186
      }
      if (accept) {
        cur ^= 1;
        fl = F_ACTIVE;
        cost_now = new_cost;
        mu = lm_relax(p, mu);
        if (it0 > 0 && is_converged(p, prev_cost0, new_cost)) {
          status = 1;  // ilqr.hh:82-84
          fl = 0;
        } else if (!((double)(it0 + 1) < p.max_iters)) {
          status = 2;  // ilqr.hh:86
          fl = 0;
        }
      } else if (trial0 + 1 >= p.ls_max_iters) {
        if (lm_restart(p, mu)) {
          restart = true;  // same iterate, larger mu: the recursion runs again
          fl = F_ACTIVE;
        } else {
          status = 3;  // ilqr.hh:191-193
          fl = 0;
        }
      }
      count_active = (fl & F_ACTIVE) != 0;
    } else if (fl == F_ACTIVE) {
      count_active = true;
    } else {
      return;  // both waves
    }
  }
  const bool run = force || !settle || ((accept || restart) && fl != 0);
  const int iters_now = (settle && accept) ? it0 + 1 : it0;
  __syncthreads();
  if (role == 1 && lane == 0) {
    if (settle) {
      if (p.mu_init > 0.0) st.mu[b] = mu;
      st.n_fwd[b] += 1;
      store_settled(st, b, accept, cur, new_cost, it0, trial0, alpha0, p.step_update, status, fl);
    }
    if (count_active) atomicAdd(active_counter(st), 1);
  }
  if (!run) return;  // back-tracking continues with the old gains, or the trajectory is done

  const int j = lane & 15, kk = lane >> 4;
  const RecLayout L = st.layout;
  const S *lin = (const S *)st.lin[cur] + rec_base(L, b, n);  // (tiled records: the host sets L.tiled = 1 when it launches this kernel)
  S *gains = (S *)st.gains + knot_base<true>(b, n, 52);
  __shared__ double ring[3][BW2_BUF];
  __shared__ double kf[2][80];  // [0..63] K, column j at [4 j ..]; [64..73] l10 l20 l30 l21 l31 l32 1/d0..1/d3
  // operand offsets inside a ring slot: record entries, or entries of the constant table behind the record
  int off[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int src;
    if (k < 3) src = m_source_tab(4 * k + kk, j);
    else if (k < 6) src = (j < 12) ? cxx_source_tab(L, 4 * (k - 3) + kk, j) : -1 - CTAB_ZERO;
    else src = L.off_g + j;
    off[k] = (src >= 0) ? src : BW2_REC + (-1 - src);
  }
  for (int t = threadIdx.x; t < CTAB_SIZE; t += 128) {
    const double v = (double)((const S *)st.ctab)[t];
    ring[0][BW2_REC + t] = v;
    ring[1][BW2_REC + t] = v;
    ring[2][BW2_REC + t] = v;
  }
  // a record is stride / 2 entry pairs, TILE2 elements apart (tiled placement): lane l fetches pair l (clamped: the lanes
  // beyond the record fetch its last pair again and drop it into ring entries nobody reads)
  typedef typename GA<S>::v2 rv2;
  typedef typename GA<S>::cptr2 rptr2;
  const int pair = (lane < L.stride / 2) ? lane : L.stride / 2 - 1;

  if (role == 1) {
    // ------------------------------------------------------------------ G: records + gradient
    auto rec_pair = [&](int i) -> rv2 { return *(rptr2)(lin + rec_elem(L, i, 2 * pair)); };
    rv2 r = {0, 0};
    {
      const rv2 a = rec_pair(n - 1);
      ring[(n - 1) % 3][2 * lane] = (double)a.x;
      ring[(n - 1) % 3][2 * lane + 1] = (double)a.y;
      if (n >= 2) {
        const rv2 b2 = rec_pair(n - 2);
        ring[(n - 2) % 3][2 * lane] = (double)b2.x;
        ring[(n - 2) % 3][2 * lane + 1] = (double)b2.y;
      }
      if (n >= 3) r = rec_pair(n - 3);
    }
    __syncthreads();
    double vxl[3] = {0.0, 0.0, 0.0};  // V_x[4 kc + kk]
    double QuTk = 0.0;
    typedef typename GA<S>::v2 sv2;
    typedef typename GA<S>::ptr2 gptr2;
    gptr2 kdst = (gptr2)(gains + knot_elem<true>(n - 1, 0, 52));  // k of knot n-1; lanes other than 0 use the dump slot
    const long kstep = (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2;
    gptr2 kdst0 = (lane == 0) ? kdst : (gptr2)((S *)st.dump + 4 * (long)b);
    gptr2 kdst1 = (lane == 0) ? kdst + TILE : (gptr2)((S *)st.dump + 4 * (long)b + 2);
    const long kst = (lane == 0) ? kstep : 0;
    auto gradient_step = [&](int q) {
      const double *buf = ring[q % 3];
      const double *f = kf[q & 1];
      const double m0 = buf[off[0]], m1 = buf[off[1]], m2 = buf[off[2]], gcj = buf[off[6]];
      double part = m0 * vxl[0] + m1 * vxl[1] + m2 * vxl[2];
      part = xor16_sum(part);
      part = xor32_sum(part);
      const double ghat = gcj + part;  // [Q_x ; Q_u][j]
      const double Qu0 = row_bcast<12>(ghat), Qu1 = row_bcast<13>(ghat), Qu2 = row_bcast<14>(ghat),
                   Qu3 = row_bcast<15>(ghat);
      const double l10 = f[64], l20 = f[65], l30 = f[66], l21 = f[67], l31 = f[68], l32 = f[69], i0 = f[70],
                   i1 = f[71], i2 = f[72], i3 = f[73];
      double kff[4];
      ldlt4_solve_neg(Ldlt4{l10, l20, l30, l21, l31, l32, i0, i1, i2, i3}, Qu0, Qu1, Qu2, Qu3, kff);
      const double k0 = kff[0], k1 = kff[1], k2 = kff[2], k3 = kff[3];  // feed-forward (ilqr.hh:128), in every lane
      const sv2 w0 = {(S)k0, (S)k1}, w1 = {(S)k2, (S)k3};
      *kdst0 = w0;
      *kdst1 = w1;
      kdst0 -= kst;
      kdst1 -= kst;
      const double c0 = f[4 * j], c1 = f[4 * j + 1], c2 = f[4 * j + 2], c3 = f[4 * j + 3];  // K[:, j]
      const double vx = ghat + (c0 * Qu0 + c1 * Qu1 + c2 * Qu2 + c3 * Qu3);  // V_x = Q_x + K^T Q_u
      QuTk += Qu0 * k0 + Qu1 * k1 + Qu2 * k2 + Qu3 * k3;
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) vxl[kc] = __shfl(vx, 4 * kc + kk);
    };
    for (int i = n - 1; i >= 0; --i) {
      if (i + 1 <= n - 1) gradient_step(i + 1);
      if (i - 2 >= 0) {
        ring[(i - 2) % 3][2 * lane] = (double)r.x;
        ring[(i - 2) % 3][2 * lane + 1] = (double)r.y;
      }
      if (i - 3 >= 0) r = rec_pair(i - 3);
      __syncthreads();
    }
    gradient_step(0);
    if (lane == 0) {
      st.terms[2 * b] = QuTk;
      st.terms[2 * b + 1] = -QuTk;  // k^T Quu k = -Q_u^T k for the exact solve (see k_backward)
      st.n_bwd[b] += 1;
      if (!force) {
        arm_line_search(p, st, b, iters_now, cost_now, QuTk, -QuTk);
      }
    }
    return;
  }

  // -------------------------------------------------------------------- M: matrix recursion
  const bool gowner = (kk == 0 && j < 12);
  const int ge0 = 4 + 4 * j;
  typedef typename GA<S>::ptr2 gptr2;
  typedef typename GA<S>::v2 sv2;
  gptr2 gdst0 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0, 52) : (S *)st.dump + 4 * (long)b);
  gptr2 gdst1 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0 + 2, 52) : (S *)st.dump + 4 * (long)b + 2);
  const long gstep = gowner ? (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2 : 0;
  // register 3 <-> row 12 + kk: C_uu = 2 R (+ mu on the diagonal, lm_restart)
  const double cuu = (j >= 12) ? 2.0 * c.R[kk * 4 + (j - 12)] + ((j - 12 == kk) ? mu : 0.0) : 0.0;
  double va[3] = {0.0, 0.0, 0.0};  // V_xx[j][4 kc + kk]  (A operand)
  __syncthreads();  // ring[(n-1) % 3], ring[(n-2) % 3] and the constant tables are filled
  double m[3], cx[3];
  {
    const double *buf = ring[(n - 1) % 3];
    m[0] = buf[off[0]]; m[1] = buf[off[1]]; m[2] = buf[off[2]];
    cx[0] = buf[off[3]]; cx[1] = buf[off[4]]; cx[2] = buf[off[5]];
  }
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  for (int i = n - 1; i >= 0; --i) {
    // operands of knot i-1, for the next iteration (the slot was filled during the previous interval)
    const double *nb = ring[(i > 0 ? i - 1 : 0) % 3];
    const double m_n0 = nb[off[0]], m_n1 = nb[off[1]], m_n2 = nb[off[2]], cx_n0 = nb[off[3]], cx_n1 = nb[off[4]],
                 cx_n2 = nb[off[5]];
    QSTAMP(0);  // operand reads issued
    const d4 T = bw_tile_T(va, m);
    QKEEP(T[0]); QKEEP(T[3]);
    QSTAMP(1);  // T = V M
    d4 H = bw_tile_H(m, T, cx, cuu);
    QKEEP(H[0]); QKEEP(H[3]);
    QSTAMP(2);  // H
    double Quu[16], Qu_unused[4], col[4];
    gather_rows(H[3], col);
    bcast_quu_row<0>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<1>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<2>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<3>(col, 0.0, Quu, Qu_unused);
    QKEEP(Quu[15]); QKEEP(Quu[0]); QKEEP(col[3]);
    QSTAMP(4);  // gather + broadcast of Q_uu
    const Ldlt4 f4 = ldlt4_factor(Quu);  // (ilqr.hh:126; unpivoted: see ldlt4_factor)
    double kcol[4];
    ldlt4_solve_neg(f4, col[0], col[1], col[2], col[3], kcol);  // K[:, j] (ilqr.hh:127)
    QKEEP(kcol[0]); QKEEP(kcol[3]);
    QSTAMP(5);  // factorisation + solve
    {
      const sv2 w0 = {(S)kcol[0], (S)kcol[1]}, w1 = {(S)kcol[2], (S)kcol[3]};
      *gdst0 = w0;
      *gdst1 = w1;
      gdst0 -= gstep;
      gdst1 -= gstep;
    }
    // hand K and the factors to G (the four lanes of a column hold the same K[:, j]: same address, same data)
    double *f = kf[i & 1];
    f[4 * j] = kcol[0]; f[4 * j + 1] = kcol[1]; f[4 * j + 2] = kcol[2]; f[4 * j + 3] = kcol[3];
    if (lane == 0) {
      f[64] = f4.l10; f[65] = f4.l20; f[66] = f4.l30; f[67] = f4.l21; f[68] = f4.l31; f[69] = f4.l32;
      f[70] = f4.i0; f[71] = f4.i1; f[72] = f4.i2; f[73] = f4.i3;
    }
    // V_xx = Q_xx + Q_xu K: A[j][kk] = Q_xu[j][kk] = H[12 + kk][j] is accumulator register 3
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(H[3], sel4(kcol, kk), H, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    m[0] = m_n0; m[1] = m_n1; m[2] = m_n2;
    cx[0] = cx_n0; cx[1] = cx_n1; cx[2] = cx_n2;
    QKEEP(va[0]); QKEEP(m[2]);
    QSTAMP(6);  // stores, hand-off to G, V_xx MFMA
    __syncthreads();
    QSTAMP(7);  // barrier
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && st.stamps)
    for (int k = 0; k < 8; ++k) st.stamps[(long)b * 8 + k] = stamp_sum[k];
#endif
}
#endif  // QILQR_WITH_BACKWARD2

}  // namespace qilqr
