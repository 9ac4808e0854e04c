// solve4.h -- k_solve4: the WHOLE of ILQR::solve (ilqr.hh:53-87) for a batch in ONE launch.
//
// A block of eight wavefronts owns four trajectories from their first linearisation to their exit status and then takes the
// next four from a queue (one atomic ticket per group), so
//   * there are no rounds: a trajectory that has converged costs nothing more, a block that is done starts other work, and
//     the step is no longer (number of rounds of the slowest trajectory) x (three launches);
//   * the host is out of the loop: no launches per round, no count of active trajectories handed back through pinned memory;
//   * gains, candidate trajectories and knot records are written and read by the same CU (L1 / this XCD's L2);
//   * the linearisation of a candidate runs BESIDE its rollout (two wavefronts follow the rollout sixteen knots behind)
//     instead of after it.
// Per group and iteration, three phases separated by block barriers; inside a phase the roles are those of the round
// kernels, through the same device functions (same arithmetic, same bits):
//   settle    waves 0..3, one trajectory each: cost of the pending candidate (left-to-right sum of its knot costs,
//             ilqr.hh:89-95), Armijo test (ilqr.hh:186), convergence tests (ilqr.hh:196-205), step-size update,
//             Levenberg-Marquardt restart -- the per-trajectory state machine of k_backward4's prologue, with the state in LDS
//   backward  waves 0..3 matrix recursions M_g, wave 4 gradients G, wave 5 record loader L (bw4_*; waves 6, 7 idle); then the expected
//             reduction test of ilqr.hh:64-68
//   forward   wave 0 control A, wave 1 pose B, wave 2 operands P (r16_*: sixteen lanes per trajectory; a SIMD each); waves 3, 6
//             and 7 linearise the candidate's knots -- dynamics blocks / cost differentials, a lane per (trajectory, knot) --
//             in chunks of sixteen as A and B announce them stored
// A trajectory that is back-tracking skips the backward phase, one that has converged skips both; the block leaves the
// loop when its four trajectories have an exit status.
#pragma once

namespace qilqr {

struct Solve4Slot {  // state of one trajectory of the block (LDS)
  int b;             // global trajectory index, -1: the slot is empty (batch not a multiple of four)
  int fl, cur, iters, trial, status, n_bwd, n_fwd;
  int run;           // this iteration's backward phase includes the trajectory
  double cost, prev_cost, alpha, mu, t0, t1;
};

// LDS of the block, at namespace scope so that the role wrappers below -- separate functions, each with its own register
// allocation -- address it as LDS (a pointer handed to a non-inlined function would be generic: flat loads).  Inlined into
// one kernel body the seven roles fight for 256 registers and the rollout's control wave spills 23 values per knot.
__shared__ __attribute__((aligned(16))) double s4_ring[4][4][BW2_BUF];
__shared__ double s4_kf[4][2][80];
__shared__ R16Lds s4_r16;
__shared__ Solve4Slot s4_slot[4];
__shared__ int s4_task;  // next (chunk, half) of the candidate's linearisation (forward phase)

template <typename S>
__device__ __attribute__((noinline)) double s4_gradient(const RecLayout &L, S *gains, S *dump4, bool grun, int n, int lane) {
  return bw4_gradient_wave<S>(s4_ring, s4_kf, L, gains, dump4, grun, n, lane);
}
template <typename S>
__device__ __attribute__((noinline)) void s4_loader(const RecLayout &L, const S *r0, const S *r1, const S *r2, const S *r3, int n, int lane) {
  bw4_loader_wave<S>(s4_ring, L, r0, r1, r2, r3, n, lane);
}
template <typename S>
__device__ __attribute__((noinline)) void s4_matrix(const RecLayout &L, int w, bool run, S *gains, S *dump4, double cuu, int n, int lane) {
  bw4_matrix_wave<S>(s4_ring, s4_kf, L, w, run, gains, dump4, cuu, n, lane, nullptr);
}
template <typename S>
__device__ __attribute__((noinline)) void s4_operands(const S *traj, const S *gains, S *out, double alpha, bool live, int n, int lane) {
  r16_wave_P<S>(s4_r16, traj, gains, out, alpha, live, n, lane, nullptr);
}
template <typename S>
__device__ __attribute__((noinline)) void s4_steps(const ModelConsts<double> &c, int parity, const S *traj, S *out, bool live, int n, int lane) {
  using namespace r16;
  auto ld0 = [&](int e) -> double { return e >= 0 ? (double)traj[knot_elem<true>(0, e, 18)] : 0.0; };
  const double TT = ld0(tt_elem(lane)), QQ = ld0(qq_elem(lane)), VL = ld0(vl_elem(lane)), VW = ld0(vw_elem(lane));
  R16_LOADS_DONE();
  r16_wave_X<S, 1>(s4_r16, c, parity, TT, QQ, VL, VW, out, live, n, lane, nullptr);
}

// one lane's share of a linearisation: the dynamics blocks (half = 0) or the cost differentials and the knot cost (half = 1)
// of knot i of trajectory b in buffer `buf` (k_linearize's body)
template <typename S, int LK>
__device__ __forceinline__ void solve4_linearize_lane(const ModelConsts<S> &cl, const S *qr, const BatchState &st, int b, int i, int n,
                                                      int buf, int half) {
  S pt[18];
  load_knot<true>((const S *)st.traj[buf] + knot_base<true>(b, n, 18), i, 18, pt);
  const TiledRecWriter<S> w{(S *)st.lin[buf] + rec_base(st.layout, b, n) + rec_elem(st.layout, i, 0)};  // (tiled records: what bw4_loader_wave reads)
  if (half == 0) {
    linearize_dynamics(cl, pt, w);
    w.flush();
    return;
  }
  S pd[18];
  if (st.desired_tiled) load_knot<true>((const S *)st.desired + knot_base<true>(b, n, 18), i, 18, pd);
  else load_knot<false>((const S *)st.desired, i, 18, pd);
  const S cost = linearize_cost<LK>(qr, qr + 144, pt, pd, w);
  w.flush();
  st.knot_cost[buf][cost_index(b, i, n)] = (double)cost;  // summed in fp64 (settle)
}

// the candidate's linearisation, sixteen knots behind the rollout (forward phase, the follower waves): tasks (chunk of
// sixteen knots, half) are taken from a counter in LDS in the order the rollout produces them -- half 0 the dynamics blocks,
// half 1 the cost differentials and knot costs; lane = (row, knot of the chunk).  A lane's share is a long serial computation
// (~35 000 cycles): the phase ends one task after the rollout, and three waves keep up with it where two do not.
template <typename S, int LK>
__device__ __attribute__((noinline)) void s4_follow(const ModelConsts<S> *cp, const S *qr, const BatchState &st, int b, int buf, bool live,
                                                    int n, int lane) {
  const int ntasks = 2 * ((n + R16_CHUNK - 1) / R16_CHUNK);
  for (;;) {
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&s4_task, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= ntasks) break;
    const int k0 = (t >> 1) * R16_CHUNK, half = t & 1;
    const int need = (k0 + R16_CHUNK < n) ? k0 + R16_CHUNK : n;
    // knots [0, need) are stored when each step wave has announced its last knot below `need`
    const int m0 = (need - 1) - ((need - 1) & 1), m1 = (need - 1) - (((need - 1) & 1) ^ 1);
    if (!(r16_flag_wait_relaxed(s4_r16, R16_F_K0, m0 + 1, lane) && (m1 < 0 || r16_flag_wait_relaxed(s4_r16, R16_F_K1, m1 + 1, lane)))) break;
    const int i = k0 + (lane & 15);
    if (live && i < n) solve4_linearize_lane<S, LK>(*cp, qr, st, b, i, n, buf, half);
  }
}
constexpr int S4_WAVES = 8, S4_THREADS = 64 * S4_WAVES;
// forward-phase roles by wave (waves w and w + 4 share SIMD w % 4): the control wave A and the pose wave B have a SIMD each;
// the operand wave P shares its SIMD with one follower, the other two followers share the fourth
constexpr int S4_W_X0 = 0, S4_W_X1 = 1, S4_W_P = 2;
__device__ __forceinline__ bool s4_is_follower(int w) { return w == 3 || w == 7 || w == 6; }
// first linearisation of the four trajectories of a group (ilqr.hh:56 needs their cost; the first backward pass their
// records): every lane of the block takes (trajectory, knot) pairs, dynamics halves first, then cost halves
template <typename S, int LK>
__device__ __attribute__((noinline)) void s4_first(const ModelConsts<S> *cp, const S *qr, const BatchState &st, int b0, int B, int n) {
  const int per = 4 * n;
  for (int t = threadIdx.x; t < 2 * per; t += S4_THREADS) {
    const int half = t >= per, r = half ? t - per : t;
    const int g = r / n, i = r - g * n;
    if (b0 + g < B) solve4_linearize_lane<S, LK>(*cp, qr, st, b0 + g, i, n, 0, half);
  }
}

template <typename S, int LK>
__global__ __launch_bounds__(S4_THREADS) void k_solve4(ModelConsts<double> c, const ModelConsts<S> *__restrict__ cp, SolveParams p, BatchState st,
                                                int B, int n, unsigned ticket_base) {
  using namespace r16;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double (&ring)[4][4][BW2_BUF] = s4_ring;
  R16Lds &rsh = s4_r16;
  Solve4Slot (&sl)[4] = s4_slot;
  __shared__ S qr[160];  // the weights Q (144) and R (16) in storage precision, for the cost half of the linearisation
  __shared__ int s_group;
  const RecLayout L = st.layout;
  for (int k = threadIdx.x; k < 160; k += S4_THREADS) qr[k] = (k < 144) ? cp->Q[k] : cp->R[k - 144];
  bw4_fill_ctab<S>(ring, st.ctab, S4_THREADS);
  if (threadIdx.x < 64) r16_exp2_lds[threadIdx.x >> 4][threadIdx.x & 15] = Series<double>::exp2[threadIdx.x >> 4][threadIdx.x & 15];  // (rollout16_body.inc)
  const int ngroups = (B + 3) / 4;
#ifdef QILQR_STAMPS
  // diagnostic build: cycles of wave `w` per phase, summed over the block's life: [0] first linearisation, [1] settle,
  // [2] backward, [3] forward (the wave's own role), [4] forward, waiting at the phase's closing barrier, [5] iterations
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif

  for (;;) {
    // ------------------------------------------------------------------------------ next group of four trajectories
    __syncthreads();  // (the previous group's last reads of sl[] are done)
    if (threadIdx.x == 0) s_group = (int)(atomicAdd((unsigned *)&st.counters[0], 1u) - ticket_base);
    __syncthreads();
    const int group = s_group;
    if (group < 0 || group >= ngroups) return;  // block-uniform
    const int b0 = group * 4;
    s4_first<S, LK>(cp, qr, st, b0, B, n);
    __syncthreads();
    QSTAMP(0);
    if (w < 4 && lane == 0) {
      Solve4Slot &s = sl[w];
      s.b = (b0 + w < B) ? b0 + w : -1;
      s.cur = 0; s.iters = 0; s.trial = 0; s.n_bwd = 0; s.n_fwd = 0; s.run = 0;
      s.status = 2;  // QILQR_STATUS_MAX_ITERS unless an exit path fires
      s.alpha = 1.0; s.mu = 0.0; s.t0 = 0.0; s.t1 = 0.0; s.cost = 0.0; s.prev_cost = 0.0;
      s.fl = (s.b >= 0 && 0.0 < p.max_iters) ? F_ACTIVE : 0;
    }
    bool first = true;  // the first settle step sums the cost of the initial trajectory instead of a candidate's

    for (;;) {
      // ---------------------------------------------------------------------------- settle (waves 0..3)
      __syncthreads();
      if (w < 4 && sl[w].b >= 0) {
        Solve4Slot &s = sl[w];
        const int b = s.b;
        int fl = s.fl, cur = s.cur;
        const bool pending = (fl & F_SEARCH) != 0;
        if (first || pending) {
          // left to right (ilqr.hh:89-95): 64 lanes fetch 64 knot costs, every lane adds them in order from broadcast
          // reads of the wave's own ring slot (nobody else touches it between the phases)
          const double *kc = st.knot_cost[first ? 0 : (cur ^ 1)];
          double *scr = &ring[w][0][0];
          double sum = 0.0;
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
              for (int e = 0; e < 8; ++e) sum += x[e];
            }
            for (; t < cnt; ++t) sum += scr[t];
          }
          if (first) {
            if (lane == 0) { s.cost = sum; s.prev_cost = sum; }
          } else {
            // acceptance of the pending candidate (ilqr.hh:70-84, 174-194)
            const int it0 = s.iters, trial0 = s.trial;
            const double prev_cost0 = s.prev_cost, alpha0 = s.alpha;
            double mu = s.mu;
            bool accept;
            if (it0 == 0) {
              accept = true;  // ilqr.hh:71-73: the first rollout is taken unconditionally
            } else {
              const double desired = p.reduction_frac * cost_reduction(s.t0, s.t1, alpha0);
              accept = (sum - prev_cost0 < desired);  // ilqr.hh:186
            }
            int status = -1;
            if (accept) {
              cur ^= 1;
              fl = F_ACTIVE;
              mu = lm_relax(p, mu);
              if (it0 > 0 && is_converged(p, prev_cost0, sum)) {
                status = 1;  // ilqr.hh:82-84
                fl = 0;
              } else if (!((double)(it0 + 1) < p.max_iters)) {
                status = 2;  // ilqr.hh:86
                fl = 0;
              }
            } else if (trial0 + 1 >= p.ls_max_iters) {
              if (lm_restart(p, mu)) {
                fl = F_ACTIVE;  // same iterate, larger mu: the recursion runs again
              } else {
                status = 3;  // ilqr.hh:191-193
                fl = 0;
              }
            }
            if (lane == 0) {
              s.mu = mu;
              s.n_fwd += 1;
              if (accept) {
                s.cur = cur;
                s.cost = sum;
                if (st.cost_hist && it0 < st.hist_cap) st.cost_hist[(long)b * st.hist_cap + it0] = sum;
                s.iters = it0 + 1;
              } else {
                s.trial = trial0 + 1;
                s.alpha = alpha0 * p.step_update;  // ilqr.hh:189
              }
              if (status >= 0) s.status = status;
              s.fl = fl;
            }
          }
        }
        if (lane == 0) s.run = (fl == F_ACTIVE) ? 1 : 0;  // accepted and continuing, restarted, or not yet started
      }
      first = false;
      __syncthreads();
      QSTAMP(1);
      const int any_fl = sl[0].fl | sl[1].fl | sl[2].fl | sl[3].fl;
      if (any_fl == 0) break;  // block-uniform: the four trajectories have their exit status

      // ---------------------------------------------------------------------------- backward
      if ((sl[0].run | sl[1].run | sl[2].run | sl[3].run) != 0) {  // block-uniform
        if (w == 4) {
          const int g = lane >> 4, j = lane & 15;
          const int bg = sl[g].b >= 0 ? sl[g].b : sl[0].b;  // a valid stand-in for an empty slot (never stored)
          const bool grun = sl[g].run != 0;
          const double QuTk = s4_gradient<S>(L, (S *)st.gains + knot_base<true>(bg, n, 52), (S *)st.dump + 4 * (long)bg, grun, n, lane);
          if (j == 0 && grun) {
            Solve4Slot &s = sl[g];
            s.t0 = QuTk;
            s.t1 = -QuTk;  // k^T Quu k = -Q_u^T k for the exact solve (see k_backward)
            s.n_bwd += 1;
            s.prev_cost = s.cost;  // ilqr.hh:61
            if (s.iters > 0 && is_converged(p, s.cost, s.cost + cost_reduction(QuTk, -QuTk, 1.0))) {
              s.status = 0;  // ilqr.hh:66-68
              s.fl = 0;
            } else if (s.iters > 0 && p.ls_max_iters <= 0) {
              s.status = 3;  // line_search with no trial allowed throws at once
              s.fl = 0;
            } else {
              s.alpha = 1.0;
              s.trial = 0;
              s.fl = F_ACTIVE | F_SEARCH;
            }
          }
        } else if (w == 5) {
          // (a trajectory that skips this backward phase is streamed as a duplicate of the first one that does not)
          const int first = sl[0].run ? 0 : (sl[1].run ? 1 : (sl[2].run ? 2 : 3));
          const S *rec[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int gs = sl[g].run ? g : first;
            rec[g] = (const S *)st.lin[sl[gs].cur] + rec_base(L, sl[gs].b, n);
          }
          s4_loader<S>(L, rec[0], rec[1], rec[2], rec[3], n, lane);
        } else if (w >= 6) {
          for (int i = n; i >= 0; --i) __syncthreads();  // no role in this phase: keep the knot barriers of bw4_* company (one + n)
        } else {
          const int bg = sl[w].b >= 0 ? sl[w].b : sl[0].b;
          const int j = lane & 15, kk = lane >> 4;
          const double mu = sl[w].mu;
          const double cuu = (j >= 12) ? 2.0 * c.R[kk * 4 + (j - 12)] + ((j - 12 == kk) ? mu : 0.0) : 0.0;
          s4_matrix<S>(L, w, sl[w].run != 0, (S *)st.gains + knot_base<true>(bg, n, 52), (S *)st.dump + 4 * (long)bg, cuu, n, lane);
        }
        __syncthreads();
        QSTAMP(2);
      }

      // ---------------------------------------------------------------------------- forward
      const int search = (sl[0].fl | sl[1].fl | sl[2].fl | sl[3].fl) & F_SEARCH;
      if (search) {  // block-uniform
        if (threadIdx.x < R16_NFLAGS) rsh.flags[threadIdx.x] = 0;
        if (threadIdx.x == 0) s4_task = 0;
        __syncthreads();
        const int row = lane >> 4;
        const bool live = (sl[row].fl & F_SEARCH) != 0;
        const unsigned long long livemask = __ballot(live);
        const int lrow = live ? row : ((__ffsll((long long)livemask) - 1) >> 4);  // rows not searching alias the first one that is
        const int bs = sl[lrow].b, cur = sl[lrow].cur;
        const S *traj = (const S *)st.traj[cur] + knot_base<true>(bs, n, 18);
        const S *gains = (const S *)st.gains + knot_base<true>(bs, n, 52);
        S *out = (S *)st.traj[cur ^ 1] + knot_base<true>(bs, n, 18);
        // a wavefront per SIMD for the three rollout roles (waves w, w + 4 share SIMD w % 4); the two linearising waves share the fourth
        if (w == S4_W_X0 || w == S4_W_X1) s4_steps<S>(c, w, traj, out, live, n, lane);
        else if (w == S4_W_P) s4_operands<S>(traj, gains, out, sl[lrow].alpha, live, n, lane);
        else if (s4_is_follower(w)) s4_follow<S, LK>(cp, qr, st, sl[row].b, sl[row].cur ^ 1, live, n, lane);
        QSTAMP(3);
        __syncthreads();
        QSTAMP(4);
#ifdef QILQR_STAMPS
        stamp_sum[5] += 1;
#endif
        if (rsh.flags[R16_F_ABORT]) {  // a hand-off never arrived (cannot happen; every spin is bounded): give up, visibly
          if (w < 4 && lane == 0 && sl[w].b >= 0) { sl[w].status = 3; sl[w].fl = 0; }
        }
      }
    }

    // ------------------------------------------------------------------------------ results of the group
    if (w < 4 && lane == 0 && sl[w].b >= 0) {
      const Solve4Slot &s = sl[w];
      const int b = s.b;
      st.cur[b] = s.cur;
      st.cost[b] = s.cost;
      st.status[b] = s.status;
      st.iters[b] = s.iters;
      st.n_bwd[b] = s.n_bwd;
      st.n_fwd[b] = s.n_fwd;
      st.flags[b] = 0;
      st.prev_cost[b] = s.prev_cost;
      st.terms[2 * b] = s.t0;
      st.terms[2 * b + 1] = s.t1;
      st.alpha[b] = s.alpha;
      st.trial[b] = s.trial;
      if (p.mu_init > 0.0) st.mu[b] = s.mu;
    }
#ifdef QILQR_STAMPS
    if (lane == 0 && st.stamps && blockIdx.x * S4_WAVES + w < B)  // (B x 8 words are allocated)
      for (int k = 0; k < 8; ++k) st.stamps[((long)blockIdx.x * S4_WAVES + w) * 8 + k] = stamp_sum[k];
#endif
  }
}

}  // namespace qilqr
