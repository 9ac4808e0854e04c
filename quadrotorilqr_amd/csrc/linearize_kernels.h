// linearize_kernels.h -- k_linearize (dynamics Jacobian blocks and cost differentials of every knot: quadrotor_model.cc:33-49, cost.hh:36-61),
// k_begin and k_init (a batch's first cost sum and the arming of its state machines: ilqr.hh:55, 89-95).
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "kernels_common.h"

namespace qilqr {

// ---------------------------------------------------------------------------------------------
// k_linearize: two threads per (b, i), in different wavefronts: one writes the dynamics Jacobian
// blocks of the knot record, the other the cost differentials and the knot cost (the kernel is bound
// by its instruction count, and the two halves share nothing but the knot they read).
// which = 0: trajectory traj[cur[b]], 1: candidate traj[cur[b]^1].
// need_flag: only problems whose flags contain it (0 = all).  round >= 0: publish the active count.
// LK: layout kind of the records (se3_math.h, layout_kind).
// ---------------------------------------------------------------------------------------------
#ifndef QILQR_LIN_BLOCK
#define QILQR_LIN_BLOCK 128
#endif
#ifndef QILQR_LIN_WAVES
#define QILQR_LIN_WAVES 3  // register budget of k_linearize in waves per SIMD: no spills (with 4, and the records' paired stores, 200-300 bytes of scratch per lane: 23.9 against 21.6 us per launch with every trajectory live, -1.7 % of a solve at B = 1024)
#endif
template <typename S, int LK, int INTEG, bool TILED>
__global__ __launch_bounds__(QILQR_LIN_BLOCK) __attribute__((amdgpu_waves_per_eu(QILQR_LIN_WAVES, QILQR_LIN_WAVES))) void
k_linearize(ModelConsts<S> c, const ModelConsts<S> *__restrict__ cp, BatchState st, int B, int n, int which,
            int need_flag, int round) {
  // The weights Q (144) and R (16) are more constants than a wave has scalar registers: the block keeps
  // them in LDS (filled from the device copy *cp) and the cost half reads them row by row where it uses
  // them; everything else comes from the by-value copy c.
  // (one copy per wavefront, filled by the wavefronts of the cost half only and without a block barrier: the
  // dynamics half does not wait for weights it never reads)
  __shared__ S qr_all[QILQR_LIN_BLOCK / 64][160];
  S *qr = qr_all[threadIdx.x >> 6];
  // thread -> (half, tile, knot, lane): the 64 lanes of a wavefront hold one knot of 64 consecutive trajectories
  long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
#ifdef QILQR_STAMPS
  unsigned long long lin_t0, lin_r0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(lin_t0), "=s"(lin_r0)::"memory");
  const long lin_wave = id >> 6;
  auto lin_stamp = [&](int half, double keep) {
    unsigned long long t1, r1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(keep) : "memory");
    if ((id & 63) == 0 && st.stamps && lin_wave < 2048 && round < 0) {  // not inside a solve: k_backward's stamps stay
      st.stamps[lin_wave * 4 + 0] = lin_r0;
      st.stamps[lin_wave * 4 + 1] = r1;
      st.stamps[lin_wave * 4 + 2] = t1 - lin_t0;
      st.stamps[lin_wave * 4 + 3] = half;
    }
  };
#endif
  if (id < COUNT_STRIPES) {  // first wavefront of block 0 (COUNT_STRIPES == 64)
    // hand the count of trajectories still active after this round's k_backward to the host: one
    // system-scope store into pinned memory, tagged with the round (no copy kernel, no event on the stream)
    int act = st.counters[COUNT_BASE + id];
    st.counters[COUNT_BASE + id] = 0;  // the next k_backward counts again
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) act += __shfl_xor(act, off);
    if (id == 0 && round >= 0)
      __hip_atomic_store(&st.host_active[round & 7],
                         ((unsigned long long)(unsigned)(round + 1) << 32) | (unsigned)act, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const long per_half = (long)((B + 63) / 64) * n * 64;
  // the cost half, the longer of the two (lone-wave time 8.6 against 4.6 us), takes the first half of the grid: at 3 200
  // wavefronts for 3 072 places (B = 1024, three per SIMD) the ones that wait for a place are then short ones
  // (21.8 -> 19.2 us per launch with every trajectory live)
  const bool cost_half = id < per_half;
  if (!cost_half) id -= per_half;
  if (id >= per_half) return;  // grid padding (whole wavefronts)
  const int lane = (int)(id & 63);
  const long rest = id >> 6;
  const int i = (int)(rest % n);
  const long b = (rest / n) * 64 + lane;
  if (cost_half) {  // wave-uniform (per_half is a multiple of 64)
    const int wl = threadIdx.x & 63;
    for (int k = wl; k < 160; k += 64) qr[k] = (k < 144) ? cp->Q[k] : cp->R[k - 144];
    __builtin_amdgcn_wave_barrier();  // written and read by this wavefront only (LDS operations of a wave stay in order)
  }
  if (b >= B) return;
  // the flag and the buffer selector are requested together (one memory latency, not two, before the knot's)
  const int fl = st.flags[b];
  const int buf = st.cur[b] ^ which;
  if (need_flag && !(fl & need_flag)) return;
  S pt[18];
  load_knot<true>((const S *)st.traj[buf] + knot_base<true>(b, n, 18), i, 18, pt);
  S *rec = (S *)st.lin[buf] + rec_base(st.layout, b, n) + rec_elem(st.layout, i, 0);  // (st.layout.tiled == TILED: the host launches the matching instantiation)
  typedef typename std::conditional<TILED, TiledRecWriter<S>, PlainRecWriter<S>>::type Writer;
  if (!cost_half) {
    const Writer wd{rec};
    if (INTEG == 1) linearize_dynamics_rk4(c, pt, wd);  // the dense M of the Runge-Kutta extension
    else linearize_dynamics(c, pt, wd);
    wd.flush();
#ifdef QILQR_STAMPS
    lin_stamp(0, (double)pt[0]);
#endif
    return;
  }
  static_assert(!(TILED && INTEG == 1), "the dense records of the Runge-Kutta extension are plain");
  const Writer w{rec + (INTEG == 1 ? LIN_M_DENSE - LIN_M_BLOCKS : 0)};  // the cost entries follow M wherever it ends
  S pd[18];
  if (st.desired_tiled) load_knot<true>((const S *)st.desired + knot_base<true>(b, n, 18), i, 18, pd);
  else load_knot<false>((const S *)st.desired, i, 18, pd);
  const S cost = linearize_cost<LK>(qr, qr + 144, pt, pd, w);
  w.flush();
  st.knot_cost[buf][cost_index(b, i, n)] = (double)cost;  // summed in fp64 (k_init / k_backward)
#ifdef QILQR_STAMPS
  lin_stamp(1, (double)cost);
#endif
}

// k_begin: thread b.  A new batch starts with every selector at buffer 0 and no flags (one launch in
// place of two hipMemsetAsync, each of which is a fill kernel plus a barrier packet).
__global__ void k_begin(BatchState st, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  st.cur[b] = 0;
  st.flags[b] = 0;
}

// Levenberg-Marquardt restarts (extension; off when mu_init == 0, which is
// the reference's behaviour).  When the line search of ilqr.hh:174-194 runs out of trials the reference
// throws; with restarts on, the trajectory instead keeps its current iterate, raises mu (mu_init first,
// then x mu_factor) and repeats the backward pass with Q_uu + mu 1 in place of Q_uu everywhere -- i.e. the
// exact LQR step of the model whose control cost carries an extra (mu / 2) |du|^2 -- and searches again
// from alpha = 1.  An accepted step divides mu by mu_factor (below mu_init it returns to 0).  Past mu_max
// the status is the reference's line-search failure.  A restart is not an iteration (ilqr.hh:58 counter).
__device__ __forceinline__ bool lm_restart(const SolveParams &p, double &mu) {
  if (!(p.mu_init > 0.0)) return false;
  const double next = (mu > 0.0) ? mu * p.mu_factor : p.mu_init;
  if (!(next <= p.mu_max)) return false;
  mu = next;
  return true;
}
__device__ __forceinline__ double lm_relax(const SolveParams &p, double mu) {
  if (!(mu > 0.0)) return mu;
  const double next = mu / p.mu_factor;
  return (next < p.mu_init) ? 0.0 : next;
}

// ---------------------------------------------------------------------------------------------
// k_init: thread b.  cost = sum of knot costs (left to right, ilqr.hh:89-95); arm the state machine.
// ---------------------------------------------------------------------------------------------
__global__ void k_init(SolveParams p, BatchState st, int B, int n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double *kc = st.knot_cost[st.cur[b]];
  double s = 0.0;
  // the additions stay in knot order; the loads are requested eight at a time
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    double v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = kc[cost_index(b, i + e, n)];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
  }
  for (; i < n; ++i) s += kc[cost_index(b, i, n)];
  st.cost[b] = s;
  st.prev_cost[b] = s;
  st.iters[b] = 0;
  st.n_bwd[b] = 0;
  st.n_fwd[b] = 0;
  st.trial[b] = 0;
  st.alpha[b] = 1.0;
  st.mu[b] = 0.0;
  st.terms[2 * b] = 0.0;
  st.terms[2 * b + 1] = 0.0;
  st.status[b] = 2;  // QILQR_STATUS_MAX_ITERS unless an exit path fires
  st.flags[b] = (0.0 < p.max_iters) ? F_ACTIVE : 0;
  if (st.orig) st.orig[b] = st.row0 + b;
  if (b == 0 && st.plan) st.plan[2] = 0;  // trajectories moved by k_compact_move in this solve (qilqr_compaction_moves)
  if (b == 0)  // both sets of counters start at zero (k_round alternates between them; a call that failed may have left counts behind)
    for (int k = 0; k < 2 * COUNT_WORDS; ++k) st.counters[k] = 0;
}

}  // namespace qilqr
