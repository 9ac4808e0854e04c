// backward1_kernel.h -- k_backward<SYM, S>: one wavefront per trajectory -- SYM = false is the GENERAL kernel (non-symmetric weights,
// force_general = 1: the reference's own forms of ilqr.hh:126-133 with Eigen's pivoted LDL^T, backward_layout.h), SYM = true the
// one-wavefront form of the symmetric recursion (force_general = 2, the Runge-Kutta extension).
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "backward_common.h"

namespace qilqr {

// SYM = true: Q and R are exactly symmetric, so V_xx and H are symmetric to rounding and the
// accumulator tile can be reused as the next knot's A operand without a transpose; no LDS and no
// barrier remain in the loop (Q_uu/Q_u are broadcast with DPP row broadcasts, the right-hand sides with
// ds_bpermute).  SYM = false: general weights, hand-offs go through padded LDS tiles.
template <bool SYM, typename S>
__global__ __launch_bounds__(64) void k_backward(ModelConsts<double> c, SolveParams p, BatchState st,
                                                 int B, int n, int force) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = threadIdx.x;
  __shared__ double cost_scr[64];  // the settle step's knot costs
  // all per-trajectory scalars are requested at once (independent loads), not one after the other
  // behind the branches that use them
  int fl = st.flags[b];
  int cur = st.cur[b];
  const int it0 = st.iters[b];
  const int trial0 = st.trial[b];
  const double prev_cost0 = st.prev_cost[b], alpha0 = st.alpha[b];
  const double term0 = st.terms[2 * b], term1 = st.terms[2 * b + 1];
  double mu = (p.mu_init > 0.0) ? st.mu[b] : 0.0;
  bool restart = false;
  if (!force) {
    if (fl & F_SEARCH) {
      // ---- acceptance of the pending candidate (ilqr.hh:70-84, 174-194), fused here so that a round
      // is three launches.  Cost = left-to-right sum of the knot costs (ilqr.hh:89-95): 64 lanes fetch
      // 64 knot costs at once, the additions stay sequential.
      const double *kc = st.knot_cost[cur ^ 1];
      double new_cost = 0.0;
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const int cnt = (n - base < 64) ? n - base : 64;
        // through LDS, every lane adding in order from broadcast reads (see k_backward4)
        cost_scr[lane] = (i < n) ? kc[cost_index(b, i, n)] : 0.0;
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
          double x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = cost_scr[t + e];
#pragma unroll
          for (int e = 0; e < 8; ++e) new_cost += x[e];
        }
        for (; t < cnt; ++t) new_cost += cost_scr[t];
      }
      const int it = it0;
      const double cost = prev_cost0;
      const double alpha = alpha0;
      bool accept;
      if (it == 0) {
        accept = true;  // ilqr.hh:71-73: the first rollout is taken unconditionally
      } else {
        const double desired = p.reduction_frac * cost_reduction(term0, term1, alpha);
        accept = (new_cost - cost < desired);  // ilqr.hh:186
      }
      int status = -1;
      if (accept) {
        cur ^= 1;
        fl = F_ACTIVE;
        mu = lm_relax(p, mu);
        if (it > 0 && is_converged(p, cost, new_cost)) {
          status = 1;  // ilqr.hh:82-84
          fl = 0;
        } else if (!((double)(it + 1) < p.max_iters)) {
          status = 2;  // ilqr.hh:86
          fl = 0;
        }
      } else {
        if (trial0 + 1 >= p.ls_max_iters) {
          if (lm_restart(p, mu)) {
            restart = true;  // same iterate, larger mu: the recursion below runs again
            fl = F_ACTIVE;
          } else {
            status = 3;  // ilqr.hh:191-193
            fl = 0;
          }
        }
      }
      if (lane == 0) {
        if (p.mu_init > 0.0) st.mu[b] = mu;
        st.n_fwd[b] += 1;
        store_settled(st, b, accept, cur, new_cost, it, trial0, alpha, p.step_update, status, fl);
        if (fl & F_ACTIVE) atomicAdd(active_counter(st), 1);
      }
      if ((!accept && !restart) || fl == 0) return;  // back-tracking continues with the old gains, or the trajectory is done
    } else if (fl == F_ACTIVE) {
      if (lane == 0) atomicAdd(active_counter(st), 1);
    } else {
      return;
    }
  }
  const int j = lane & 15, kk = lane >> 4;
  const RecLayout L = st.layout;
  // the recursion itself is always fp64 (fp64 MFMA); S is only the type of the records read and of
  // the gains written
  const S *lin = (const S *)st.lin[cur] + rec_base(L, b, n);  // (plain records: the host sets L.tiled = 0 when it launches this kernel)
  S *gains = (S *)st.gains + knot_base<true>(b, n, 52);

  constexpr int LD = 17;  // padded leading dimension: column reads of a row-major tile
  __shared__ double Hs[SYM ? 1 : 16 * LD];  // (general kernel: the right-hand sides of every other knot cross the tile here)

  // Seven operands per lane and knot: three elements of M = [J_x | J_u] (rows kk, 4+kk, 8+kk of
  // column j), three of C_xx (accumulator layout: register r <-> row 4 r + kk, column j) and one of
  // [C_x ; C_u].  Each is either an entry of the knot record (pointer walks back one record per knot)
  // or a constant (pointer into the constant table, step 0): the loads are unconditional.
  typename GA<S>::cptr op[7];
  long step[7];
  {
    const long knot_step = rec_elem(L, 1, 0) - rec_elem(L, 0, 0);  // one knot back
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      int src;
      if (k < 3) src = m_source_tab(L, 4 * k + kk, j);
      else if (k < 6) src = (j < 12) ? cxx_source_tab(L, 4 * (k - 3) + kk, j) : -1 - CTAB_ZERO;
      else src = L.off_g + j;
      op[k] = (typename GA<S>::cptr)((src >= 0) ? lin + rec_elem(L, n - 1, src) : (const S *)st.ctab + (-1 - src));
      step[k] = (src >= 0) ? knot_step : 0;
    }
  }
  // The general kernel alternates between two kinds of knot (round 5; below): the odd kind takes C TRANSPOSED -- entry [j][4 r + kk] where the
  // even kind takes [4 r + kk][j] (the same entry when the record stores a symmetric C_xx).
  typename GA<S>::cptr opt[3];
  long stept[3];
  if constexpr (!SYM) {
    const long knot_step = rec_elem(L, 1, 0) - rec_elem(L, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int src = (j < 12) ? cxx_source_tab(L, j, 4 * k + kk) : -1 - CTAB_ZERO;
      opt[k] = (typename GA<S>::cptr)((src >= 0) ? lin + rec_elem(L, n - 1, src) : (const S *)st.ctab + (-1 - src));
      stept[k] = (src >= 0) ? knot_step : 0;
    }
  }
  // gain slots of this lane for knot n-1, walked back one knot per iteration (tiled layout: one
  // 16-byte slot per element pair); lanes that own nothing point at the dump slot with step 0
  const bool gowner = (kk == 0 && j <= 12);
  const int ge0 = (j < 12) ? 4 + 4 * j : 0;
  typedef typename GA<S>::ptr2 gptr2;
  typedef typename GA<S>::v2 sv2;
  gptr2 gdst0 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0, 52) : (S *)st.dump + 4 * (long)b);
  gptr2 gdst1 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0 + 2, 52) : (S *)st.dump + 4 * (long)b + 2);
  const long gstep = gowner ? (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2 : 0;
  // register 3 <-> row 12 + kk: C_uu = 2 R (cost.hh:55) in columns 12..15
  const double cuu = (j >= 12) ? 2.0 * c.R[kk * 4 + (j - 12)] + ((j - 12 == kk) ? mu : 0.0) : 0.0;
  const double cuut = (j >= 12) ? 2.0 * c.R[(j - 12) * 4 + kk] + ((j - 12 == kk) ? mu : 0.0) : 0.0;  // C_uu^T (general kernel, odd knots)

  double va[3] = {0.0, 0.0, 0.0};   // V_xx[j][4 kc + kk]  (A operand)
  double vxl[3] = {0.0, 0.0, 0.0};  // V_x[4 kc + kk]
  double QuTk = 0.0, kTQuuk = 0.0;

  // software pipeline: the operands of knot i-1 are requested before the chain of knot i starts
  double m[3], cx[3], gcj;
  m[0] = (double)*op[0]; m[1] = (double)*op[1]; m[2] = (double)*op[2];
  if constexpr (SYM) {
    cx[0] = (double)*op[3]; cx[1] = (double)*op[4]; cx[2] = (double)*op[5];
  } else {
    cx[0] = (double)*opt[0]; cx[1] = (double)*opt[1]; cx[2] = (double)*opt[2];
  }
  gcj = (double)*op[6];

#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  // KIND 0: the symmetric recursion.  KIND 1 / 2: the general kernel's two kinds of knot (round 5).  An accumulator tile X used as the A
  // operand of the next product IS X^T.  The symmetric recursion lives on that (V = V^T to rounding); a non-symmetric V_xx needed a real
  // transpose through LDS every knot -- until the two kinds alternate:
  //   kind 1: the accumulator holds V.  A = V^T, so the products give T' = V^T M and, started from C^T, H' = C^T + M^T V^T M = H^T.  Rows 12..15
  //           of H^T are the COLUMNS 12..15 of H: register 3 of lane (j, a) is H[j][12 + a] = Q_xu[j][a] -- the right-hand sides, in place --
  //           and Q_uu[b][a] sits in lane (12 + b, a).  The update is formed transposed as well: V_new^T = Q_xx^T - K^T (Q_uu^T K), one matrix
  //           instruction with A[i][kk] = K[kk][i], B[kk][j] = -(K^T Q_uu)[j][kk] -- the same products and sums as the plain form.
  //   kind 2: the accumulator holds V^T.  A = V: T = V M, H = C + M^T T as written (ilqr.hh:118-124); Q_xu[j][a] = H[j][12 + a] is then a
  //           column of the tile and crosses it through LDS; V_new = Q_xx - (K^T Q_uu) K in place (ilqr.hh:133) -- and the next knot is kind 1.
  // No transpose of V_xx anywhere, one trip through LDS every other knot.  The same FORMULAS as the reference's at every knot, NOT one fixed
  // evaluation order of ilqr.hh:118-124: the triple product M^T V M is grouped (M^T V) M at kind-1 knots and M^T (V M) at kind-2 knots -- the
  // same products, summed in another grouping every other knot, equal to rounding (and, beyond ~150 knots where the unsymmetrised recursion
  // is noise in the reference too, noise of another size: tests/test_gpu_parity.py::test_long_horizon_instability...; knot-resolved bound at
  // 100 and 150 knots: test_general_kernel_pass_at_long_horizons_stays_within_the_recursion_s_own_sensitivity).
  auto knot = [&](int i, auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
    if (i > 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) op[k] -= step[k];
      if constexpr (!SYM) {
#pragma unroll
        for (int k = 0; k < 3; ++k) opt[k] -= stept[k];
      }
    }
    // (loaded in storage precision, converted where first used, so that the conversion does not wait
    // on the load at the top of the loop; the next knot of the general kernel is of the other kind: C transposed behind a kind-2 knot)
    const S m_s0 = *op[0], m_s1 = *op[1], m_s2 = *op[2], g_s = *op[6];
    const S cx_s0 = (KIND == 2) ? *opt[0] : *op[3], cx_s1 = (KIND == 2) ? *opt[1] : *op[4], cx_s2 = (KIND == 2) ? *opt[2] : *op[5];
    QSTAMP(0);  // prefetch issue
    const d4 T = bw_tile_T(va, m);
    QKEEP(T[0]); QKEEP(T[3]);
    QSTAMP(1);  // T = V M (3 MFMA) complete
    d4 H = bw_tile_H(m, T, cx, (KIND == 1) ? cuut : cuu);
    QKEEP(H[0]); QKEEP(H[3]);
    QSTAMP(2);  // H (3 MFMA) complete
    // [Q_x ; Q_u] = [C_x ; C_u] + M^T V_x
    double part = m[0] * vxl[0] + m[1] * vxl[1] + m[2] * vxl[2];
    part = xor16_sum(part);
    part = xor32_sum(part);
    const double ghat = gcj + part;

    QKEEP(ghat);
    QSTAMP(3);  // gradient
    // every lane: Q_uu (4x4), Q_u; lane column j < 12: its row of Q_xu
    double Quu[16], Qu[4], rhs[4];
    if constexpr (SYM) {
      // rows 12..15 of H live in register 3: lane (j, kk) holds H[12 + kk][j].  Gather the four rows
      // of each column into every lane (permlane swaps): column j < 12 is the right-hand side
      // Q_xu[j][:] (= Q_ux[:][j] by symmetry), columns 12..15 are Q_uu, broadcast inside each row of 16 lanes
      // (lower triangle only; Q_uu is symmetric here).
      double col[4];
      gather_rows(H[3], col);
      bcast_quu_row<0>(col, ghat, Quu, Qu);
      bcast_quu_row<1>(col, ghat, Quu, Qu);
      bcast_quu_row<2>(col, ghat, Quu, Qu);
      bcast_quu_row<3>(col, ghat, Quu, Qu);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = a + 1; bb < 4; ++bb) Quu[a * 4 + bb] = Quu[bb * 4 + a];
#pragma unroll
      // lane 12: feed-forward.  Lanes 13..15 solve against a column of Q_uu itself; nobody reads them.
      for (int a = 0; a < 4; ++a) rhs[a] = (j == 12) ? Qu[a] : col[a];
    } else {
      double col[4];
      if constexpr (KIND == 1) {
        // H^T in the accumulator: col[a] in lane j = H^T[12 + a][j] = H[j][12 + a] -- Q_xu[j][a] for j < 12, Q_uu[j - 12][a] for j >= 12
        gather_rows(H[3], col);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          Quu[0 * 4 + bb] = row_bcast<12>(col[bb]); Quu[1 * 4 + bb] = row_bcast<13>(col[bb]);
          Quu[2 * 4 + bb] = row_bcast<14>(col[bb]); Quu[3 * 4 + bb] = row_bcast<15>(col[bb]);
        }
        Qu[0] = row_bcast<12>(ghat); Qu[1] = row_bcast<13>(ghat); Qu[2] = row_bcast<14>(ghat); Qu[3] = row_bcast<15>(ghat);
#pragma unroll
        for (int a = 0; a < 4; ++a) rhs[a] = (j < 12) ? col[a] : ((j == 12) ? Qu[a] : 0.0);  // lane 12: feed-forward
      } else {
        // H in the accumulator: Q_xu[j][a] = H[j][12 + a] sits in its COLUMNS 12..15 (lane (12 + a, j & 3), register j >> 2): the right-hand
        // sides cross the tile through LDS -- columns 12..15 of rows 0..11 only.  Q_uu (all sixteen entries: K^T Q_uu below is not symmetric
        // arithmetic) and Q_u come from registers while that round trip is in flight: rows 12..15 of H are register 3, gathered and broadcast
        // as in the symmetric kernels.
        if (j >= 12) {
#pragma unroll
          for (int r = 0; r < 3; ++r) Hs[(4 * r + kk) * LD + j] = H[r];
        }
        __syncthreads();
        double xr[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xr[a] = Hs[(j < 12 ? j : 0) * LD + 12 + a];
        gather_rows(H[3], col);  // col[a] in lane j = H[12 + a][j]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          Quu[a * 4 + 0] = row_bcast<12>(col[a]); Quu[a * 4 + 1] = row_bcast<13>(col[a]);
          Quu[a * 4 + 2] = row_bcast<14>(col[a]); Quu[a * 4 + 3] = row_bcast<15>(col[a]);
        }
        Qu[0] = row_bcast<12>(ghat); Qu[1] = row_bcast<13>(ghat); Qu[2] = row_bcast<14>(ghat); Qu[3] = row_bcast<15>(ghat);
#pragma unroll
        for (int a = 0; a < 4; ++a) rhs[a] = (j < 12) ? xr[a] : ((j == 12) ? Qu[a] : 0.0);  // lane 12: feed-forward
      }
    }
    QKEEP(Quu[15]); QKEEP(Quu[0]); QKEEP(Qu[3]); QKEEP(rhs[3]); QKEEP(rhs[0]);
    QSTAMP(4);  // broadcast of Q_uu, Q_u, right-hand sides
    // one right-hand side per lane: K[:, j] = -Quu^-1 Q_xu[j, :]^T in lanes j < 12 and k = -Quu^-1 Q_u in lane 12
    // (ilqr.hh:127-128); k is then broadcast
    double kcol[4];
    if constexpr (SYM) {
      // LDL^T of the lower triangle of Q_uu without pivoting (Q_uu = 2 R + J_u^T V_xx J_u is positive definite for the
      // weights this kernel is launched for; the reference's Eigen LDLT pivots on the diagonal: same result in exact arithmetic)
      const Ldlt4 f = ldlt4_factor(Quu);
      QKEEP(f.i3); QKEEP(f.l32); QKEEP(f.l31);
      ldlt4_solve_neg(f, rhs[0], rhs[1], rhs[2], rhs[3], kcol);
    } else {
      // the reference's factorisation: Eigen's diagonally pivoted LDL^T (ilqr.hh:126), restated in ldlt4_pivoted_solve
      double xs[4];
      ldlt4_pivoted_solve(Quu, rhs, xs);
      kcol[0] = -xs[0]; kcol[1] = -xs[1]; kcol[2] = -xs[2]; kcol[3] = -xs[3];
    }
    QKEEP(kcol[0]); QKEEP(kcol[3]);
    QSTAMP(5);  // factorisation + solve
    // gains of knot i: [k(4) | K column-major]; lane j < 12 owns column j, lane 12 owns k.
    // Every lane stores (lanes that own nothing write whatever they hold to a per-trajectory dump slot
    // nobody reads): no branch around the stores, so the wait for the next knot's operands is an exact
    // vmcnt(2), not vmcnt(0), and no select in front of them.
    {
      const sv2 w0 = {(S)kcol[0], (S)kcol[1]}, w1 = {(S)kcol[2], (S)kcol[3]};
      *gdst0 = w0;
      *gdst1 = w1;
      gdst0 -= gstep;
      gdst1 -= gstep;
    }
    // expected cost reduction terms (ilqr.hh:136-140): in lane 12 the right-hand side is Q_u and the
    // solution is k, so Q_u^T k = rhs . kcol there; every lane accumulates its own column's value and
    // lane 12's sum is read after the loop
    QuTk += rhs[0] * kcol[0] + rhs[1] * kcol[1] + rhs[2] * kcol[2] + rhs[3] * kcol[3];
    double vx;
    if constexpr (SYM) {
      // With Q_uu symmetric and K = -Quu^-1 Q_ux, k = -Quu^-1 Q_u, the reference's updates
      //   V_x = Q_x - K^T Quu k,  V_xx = Q_xx - K^T Quu K,  k^T Quu k      (ilqr.hh:132-133, 139)
      // are, term by term,  Q_x + K^T Q_u,  Q_xx + Q_xu K,  -Q_u^T k  (they differ from the reference's
      // evaluation by the residual of the 4x4 solve, ~ cond(Quu) eps).  That removes the product
      // K^T Quu (16 FMA per lane) from the serial chain, and the A operand of the update
      //   A[j][kk] = Q_xu[j][kk] = H[12 + kk][j]
      // is accumulator register 3 as it stands.
      vx = ghat + (kcol[0] * Qu[0] + kcol[1] * Qu[1] + kcol[2] * Qu[2] + kcol[3] * Qu[3]);
      QKEEP(vx); QKEEP(QuTk);
      QSTAMP(6);  // V_x, reduction term
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) vxl[kc] = __shfl(vx, 4 * kc + kk);  // V_x[r] lives in lanes with j == r
      H = __builtin_amdgcn_mfma_f64_16x16x4f64(H[3], sel4(kcol, kk), H, 0, 0, 0);
    } else {
      // (K^T Quu)[j][:], then V_x = Q_x - (K^T Quu) k   (ilqr.hh:132)
      double mc[4], kff[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) kff[a] = bcast_lane(kcol[a], 12);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
        mc[bb] = kcol[0] * Quu[bb] + kcol[1] * Quu[4 + bb] + kcol[2] * Quu[8 + bb] + kcol[3] * Quu[12 + bb];
      vx = ghat - (mc[0] * kff[0] + mc[1] * kff[1] + mc[2] * kff[2] + mc[3] * kff[3]);
      kTQuuk += mc[0] * kcol[0] + mc[1] * kcol[1] + mc[2] * kcol[2] + mc[3] * kcol[3];
      QKEEP(mc[3]); QKEEP(vx); QKEEP(QuTk); QKEEP(kTQuuk);
      QSTAMP(6);  // K^T Quu, V_x, reduction terms
      // V_xx = Q_xx - (K^T Quu) K   (ilqr.hh:133): one more MFMA on the same accumulator, A[j][kk] = -(K^T Quu)[j][kk], B[kk][j] = K[kk][j]
      // (kind 2); on H^T (kind 1) the transposed update V_new^T = Q_xx^T - K^T (Quu^T K): A[i][kk] = K[kk][i], B[kk][j] = -(K^T Quu)[j][kk]
      if constexpr (KIND == 1) H = __builtin_amdgcn_mfma_f64_16x16x4f64(sel4(kcol, kk), -sel4(mc, kk), H, 0, 0, 0);
      else H = __builtin_amdgcn_mfma_f64_16x16x4f64(-sel4(mc, kk), sel4(kcol, kk), H, 0, 0, 0);
    }

    // hand V_xx, V_x to the next knot
    if constexpr (SYM) {
      // V symmetric: the accumulator tile IS the next A operand.  Lanes j >= 12 hold Q_xu / Q_uu
      // leftovers there, i.e. rows 12..15 of the A operand, which only reach rows 12..15 of T
      // (register 3), and those are never used: no masking needed.
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    } else {
      // the accumulator (V or V^T) is the next A operand (V^T or V): what the next knot's kind expects; lanes j >= 12 hold leftovers that
      // only reach rows 12..15 of T, which nobody reads (as in the symmetric kernels)
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) vxl[kc] = __shfl(vx, 4 * kc + kk);  // V_x[r] lives in lanes with j == r
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    }
    m[0] = (double)m_s0; m[1] = (double)m_s1; m[2] = (double)m_s2;
    cx[0] = (double)cx_s0; cx[1] = (double)cx_s1; cx[2] = (double)cx_s2;
    gcj = (double)g_s;
    QKEEP(va[0]); QKEEP(vxl[2]);
    QSTAMP(7);  // V_xx MFMA, gain stores, hand-off
  };
  if constexpr (SYM) {
    for (int i = n - 1; i >= 0; --i) knot(i, std::integral_constant<int, 0>());
  } else {
    for (int i = n - 1; i >= 0; i -= 2) {
      knot(i, std::integral_constant<int, 1>());
      if (i >= 1) knot(i - 1, std::integral_constant<int, 2>());
    }
  }

#ifdef QILQR_STAMPS
  if (lane == 0 && st.stamps)
    for (int k = 0; k < 8; ++k) st.stamps[(long)b * 8 + k] = stamp_sum[k];
#endif
  QuTk = bcast_lane(QuTk, 12);
  kTQuuk = SYM ? -QuTk : bcast_lane(kTQuuk, 12);
  if (lane == 0) {
    st.terms[2 * b] = QuTk;
    st.terms[2 * b + 1] = kTQuuk;
    st.n_bwd[b] += 1;
    if (!force) {
      arm_line_search(p, st, b, st.iters[b], st.cost[b], QuTk, kTQuuk);
    }
  }
}

}  // namespace qilqr
