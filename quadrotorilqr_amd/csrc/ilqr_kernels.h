// ilqr_kernels.h -- device kernels of the batched iLQR solver (gfx950 only).
//
// One outer "round" of the solver is three launches, back to back on one stream:
//   k_backward4 / k_backward   settles the previous candidate (cost sum, Armijo test, convergence tests:
//                ilqr.hh:61-84, 174-194), then the Riccati recursion on the fp64 matrix core
//                (ilqr.hh:97-147): four matrix wavefronts, a gradient and a loader wavefront per four trajectories
//                (k_backward4), or one wavefront per trajectory (k_backward)
//                (k_backward2 -- a matrix and a gradient wavefront per trajectory -- and the one-launch k_solve4 are in the
//                diagnostics build: -DQILQR_DIAG, `make diag`)
//   k_rollout3 / k_rollout     closed-loop forward simulation (ilqr.hh:149-172): pose wave + control
//                wave + loader wave per 64 trajectories, or one lane per trajectory in one wavefront
//   k_linearize  two lanes per knot: dynamics Jacobian blocks / cost differentials + knot cost of the
//                candidate trajectory (quadrotor_model.cc:33-49, cost.hh:36-61); also hands the
//                count of still-active trajectories to the host (pinned memory)
// (k_accept is the stand-alone acceptance step of the qilqr_line_search entry point.)
// Every trajectory carries its own outer-iteration counter, step size and state machine, so
// trajectories that are back-tracking and trajectories that already accepted a step advance
// in the same round; the host only reads one count of still-active trajectories, late.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef QILQR_DIAG  // the diagnostics build carries the kernels that measured behind the product's, and fault injection
#define QILQR_WITH_SOLVE4 1
#define QILQR_WITH_BACKWARD2 1
#endif

#include <type_traits>

#include "backward_layout.h"
#include "rollout16.h"
#include "se3_math.h"

namespace qilqr {

// The count of still-active trajectories is kept in COUNT_STRIPES words, one per residue of the block index:
// thousands of atomic adds on ONE word are served one after the other (measured: 1024 of them, one per
// wavefront at the start of k_backward4, held every block at its first barrier for 7 us).
constexpr int COUNT_BASE = 8, COUNT_STRIPES = 64, COUNT_WORDS = COUNT_BASE + COUNT_STRIPES;
constexpr int F_ACTIVE = 1;  // still iterating
constexpr int F_SEARCH = 2;  // has gains, needs a (further) rollout trial

struct SolveParams {
  double step_update, reduction_frac, rtol, atol, max_iters;
  int ls_max_iters;
  // Levenberg-Marquardt restarts (an extension; the reference has none and mu_init = 0 switches them
  // off): see lm_restart below
  double mu_init, mu_factor, mu_max;
};

// all device pointers; [B] unless noted
struct BatchState {
  // Buffers whose element type S is the solver's storage precision (double, or float in the
  // mixed-precision mode): kernels are instantiated on S and cast.
  void *traj[2];         // TILED (se3_math.h) [tile][n][9][TILE][2]: current / candidate trajectories
  void *lin[2];          // [B][n][layout.stride] knot records of traj[k] (se3_math.h, rec_base / rec_elem)
  RecLayout layout;
  double *knot_cost[2];  // [tile][n][TILE]
  void *gains;           // TILED [tile][n][26][TILE][2]
  const void *desired;   // shared: plain [n_desired][18]; per problem: TILED like traj
  int desired_tiled;     // 0 shared, 1 per problem
  int *cur;              // which of traj[] / lin[] is current
  double *cost;          // cost of the current trajectory ("new_cost", ilqr.hh:56)
  double *prev_cost;     // "cost" inside the iteration (ilqr.hh:61)
  double *terms;         // [B][2] QuTk, kTQuuk
  double *alpha;
  double *mu;            // regularisation currently added to the diagonal of Q_uu (0 unless restarts are on)
  int *trial;
  int *flags;
  int *status, *iters, *n_bwd, *n_fwd;
  int *counters;         // [COUNT_BASE + stripe]: trajectories still active, counted by k_backward (active_counter)
  unsigned long long *host_active;  // pinned host memory, 8 words: (round + 1) << 32 | active count (k_linearize)
  unsigned long long *host_error;   // pinned host memory, one word, zero unless a kernel gave up: 1 << 32 | block (k_rollout16: a
                                    // hand-off between its wavefronts never arrived); the host turns it into QILQR_ERR_HIP
  double *cost_hist;     // [B][hist_cap] or null
  int hist_cap;
  const void *ctab;      // constant operand table (backward_layout.h)
  void *dump;            // [B][4] write-only sink for the lanes of k_backward that own no gain entry
  unsigned long long *stamps;  // diagnostic build only (-DQILQR_STAMPS): [B][8] cycle sums per section of k_backward
  // compaction of the live trajectories (k_compact_plan / k_compact_move): the caller's row of the trajectory in slot b
  // (-1: the slot's trajectory has moved away), the first row of this (sub-)batch, and the plan of the current round
  int *orig;
  int row0;
  int *plan;  // [0] moves, [16 ..] destination slots, [16 + B ..] source slots, [16 + 2 B ..] per pair: row, selectors (k_compact_plan)
};

__device__ __forceinline__ int *active_counter(const BatchState &st) {
  return st.counters + COUNT_BASE + (blockIdx.x & (COUNT_STRIPES - 1));
}

typedef double d4 __attribute__((ext_vector_type(4)));
// explicit global address space: a pointer selected between two buffers is otherwise 'generic' and
// becomes flat_load (out-of-order return, forces vmcnt(0) + lgkmcnt(0) waits)
template <typename S>
struct GA {  // global address space views of storage type S
  typedef const S __attribute__((address_space(1))) *cptr;
  typedef S v2 __attribute__((ext_vector_type(2)));
  typedef v2 __attribute__((address_space(1))) *ptr2;
  typedef const v2 __attribute__((address_space(1))) *cptr2;
};

#ifdef QILQR_STAMPS
// In-kernel section timing for a separate diagnostic build (cdna_hip_programming.md section 7): one
// s_memtime per section boundary, sums kept per wavefront, written to a buffer nothing else reads.
#define QSTAMP(slot)                                                                   \
  do {                                                                                 \
    unsigned long long _t;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    stamp_sum[slot] += _t - stamp_prev;                                                \
    stamp_prev = _t;                                                                   \
  } while (0)
#define QKEEP(x) asm volatile("" ::"v"(x))
#else
#define QSTAMP(slot) do { } while (0)
#define QKEEP(x) do { } while (0)
#endif

// Workgroups go to the eight XCDs round-robin by blockIdx, and every XCD has an L2 of its own.  The kernels that give a
// block to FOUR trajectories (k_rollout16, k_backward4) take logical block = hardware block: block g of either kernel runs
// on XCD g mod 8, so a block of k_rollout16 finds the gains in the L2 its k_backward4 block wrote them through.
// (Rounds 2-3 handed XCD x a CONTIGUOUS range of logical blocks -- introduced when tiles were 64 trajectories wide and
// sixteen blocks shared every line: FETCH_SIZE per launch 53 MB with the identity map, 31 with that one.  With tiles of four
// no two blocks share a line and the two maps measure the same at every batch size (profiles/r04_compaction.txt); the
// contiguous map is wrong for a batch whose live trajectories are a dense prefix -- k_compact_* below -- which it would
// put on the first XCDs only: a batch sorted longest-first ran its backward passes 16 % SLOWER than unsorted with it.)
__device__ __forceinline__ int xcd_local_block(unsigned hw_block, unsigned /*nblocks*/) { return (int)hw_block; }
__device__ __forceinline__ bool is_converged(const SolveParams &p, double cost, double new_cost) {
  // ilqr.hh:196-205 (cost == 0 gives NaN < rtol == false and falls through to atol)
  if (fabs(cost - new_cost) / fabs(cost) < p.rtol) return true;
  if (fabs(cost - new_cost) < p.atol) return true;
  return false;
}
__device__ __forceinline__ double cost_reduction(double QuTk, double kTQuuk, double step) {
  return step * QuTk + step * step * kTQuuk / 2.0;  // ilqr.hh:18-22
}

// ---- The stores of the per-trajectory state machine: the settle step of a candidate (ilqr.hh:70-84, 174-194) and the arming of the
// next line search behind a backward pass (ilqr.hh:61-68).
// WRITTEN WITHOUT COMPLEMENTARY BRANCHES ON PURPOSE.  `if (a) st.x[b] = u; else st.y[b] = v;` with x and y of one type is turned by
// LLVM (sinking of common code in SimplifyCFG) into ONE store through a selected address.  In the diagnosis build of rounds 3 and 4
// (kernel bodies as __device__ functions, SolveParams by const reference: the BatchState pointers then live in scratch) the AMDGPU
// backend selected that address per lane from two scratch offsets and left the wrong one in place for the lanes of the last branch:
// `st.trial[b] = 0` of the arming step became `st.status[b] = 0`, the next line search started from the previous one's trial count,
// and with one to three trials per search (Levenberg-Marquardt restarts) a solve took restarts the oracle did not -- the "flat-pointer
// anomaly" of VERDICT r03 item 8 (profiles/r04_flat_anomaly.txt has the two instruction sequences).  The product
// build compiled the same source correctly, by luck of its register allocation.  Here every word is stored unconditionally with a
// selected VALUE, or under a condition that no other store of its type complements: there is nothing for that transformation to merge.
__device__ __forceinline__ void store_settled(const BatchState &st, int b, bool accept, int cur, double new_cost, int it0, int trial0,
                                              double alpha0, double step_update, int status, int fl) {
  if (accept) {  // (cur has been flipped by the caller)
    st.cur[b] = cur;
    st.cost[b] = new_cost;
    if (st.cost_hist && it0 < st.hist_cap) st.cost_hist[(long)b * st.hist_cap + it0] = new_cost;
    st.iters[b] = it0 + 1;
  }
  st.trial[b] = accept ? trial0 : trial0 + 1;
  st.alpha[b] = accept ? alpha0 : alpha0 * step_update;  // ilqr.hh:189
  if (status >= 0) st.status[b] = status;
  st.flags[b] = fl;
}
// behind a backward pass on the trajectory's current iterate: ilqr.hh:61 (cost), :66-68 (expected reduction below the convergence
// thresholds: status 0), a line search that allows no trial throws at once (status 3), otherwise the search starts from alpha = 1
__device__ __forceinline__ void arm_line_search(const SolveParams &p, const BatchState &st, int b, int iters_now, double cost_now,
                                                double QuTk, double kTQuuk) {
  st.prev_cost[b] = cost_now;
  const bool conv = iters_now > 0 && is_converged(p, cost_now, cost_now + cost_reduction(QuTk, kTQuuk, 1.0));
  const bool none = !conv && iters_now > 0 && p.ls_max_iters <= 0;
  const bool search = !(conv || none);
  st.alpha[b] = 1.0;  // (alpha and trial are of no consequence for a trajectory that stops here)
  st.trial[b] = 0;
  if (!search) st.status[b] = conv ? 0 : 3;
  st.flags[b] = search ? (F_ACTIVE | F_SEARCH) : 0;
}

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

// ---------------------------------------------------------------------------------------------
// k_backward: one wavefront per trajectory (block = 64 threads).  See backward_layout.h.
// force = 1: run on every trajectory, no convergence test (the stand-alone backwards_pass API).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sel4(const double v[4], int kk) {
  const double lo = (kk & 1) ? v[1] : v[0], hi = (kk & 1) ? v[3] : v[2];
  return (kk & 2) ? hi : lo;
}
// 1/x to fp64 accuracy (not correctly rounded): hardware estimate + two Newton steps; half the
// dependent depth of the IEEE division sequence, which matters on the per-knot serial chain
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
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
  __shared__ double Vs[SYM ? 1 : 12 * LD];
  __shared__ double Hs[SYM ? 1 : 16 * LD];

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

  double va[3] = {0.0, 0.0, 0.0};   // V_xx[j][4 kc + kk]  (A operand)
  double vxl[3] = {0.0, 0.0, 0.0};  // V_x[4 kc + kk]
  double QuTk = 0.0, kTQuuk = 0.0;

  // software pipeline: the operands of knot i-1 are requested before the chain of knot i starts
  double m[3], cx[3], gcj;
  m[0] = (double)*op[0]; m[1] = (double)*op[1]; m[2] = (double)*op[2];
  cx[0] = (double)*op[3]; cx[1] = (double)*op[4]; cx[2] = (double)*op[5];
  gcj = (double)*op[6];

#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  for (int i = n - 1; i >= 0; --i) {
    if (i > 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) op[k] -= step[k];
    }
    // (loaded in storage precision, converted where first used, so that the conversion does not wait
    // on the load at the top of the loop)
    const S m_s0 = *op[0], m_s1 = *op[1], m_s2 = *op[2], cx_s0 = *op[3], cx_s1 = *op[4], cx_s2 = *op[5], g_s = *op[6];
    QSTAMP(0);  // prefetch issue
    const d4 T = bw_tile_T(va, m);
    QKEEP(T[0]); QKEEP(T[3]);
    QSTAMP(1);  // T = V M (3 MFMA) complete
    d4 H = bw_tile_H(m, T, cx, cuu);
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
      // Q_xu[j][a] = H[j][12 + a] sits in the accumulator's COLUMNS 12..15 (lane (12 + a, j & 3), register j >> 2): the right-hand sides
      // cross the tile through LDS -- columns 12..15 of rows 0..11 only.  Q_uu (all sixteen entries: K^T Q_uu below is not symmetric
      // arithmetic) and Q_u come from registers while that round trip is in flight: rows 12..15 of H are register 3, gathered and
      // broadcast as in the symmetric kernels (until round 5 all twenty went through LDS behind the barrier).
      if (j >= 12) {
#pragma unroll
        for (int r = 0; r < 3; ++r) Hs[(4 * r + kk) * LD + j] = H[r];
      }
      __syncthreads();
      double xr[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xr[a] = Hs[(j < 12 ? j : 0) * LD + 12 + a];
      double col[4];
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
      // V_xx = Q_xx - (K^T Quu) K   (ilqr.hh:133): one more MFMA on the same accumulator,
      // A[j][kk] = -(K^T Quu)[j][kk], B[kk][j] = K[kk][j]
      H = __builtin_amdgcn_mfma_f64_16x16x4f64(-sel4(mc, kk), sel4(kcol, kk), H, 0, 0, 0);
    }

    // hand V_xx, V_x to the next knot
    if constexpr (SYM) {
      // V symmetric: the accumulator tile IS the next A operand.  Lanes j >= 12 hold Q_xu / Q_uu
      // leftovers there, i.e. rows 12..15 of the A operand, which only reach rows 12..15 of T
      // (register 3), and those are never used: no masking needed.
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    } else {
      // accumulator layout -> A-operand layout through LDS (a transpose)
      if (j < 12) {
#pragma unroll
        for (int r = 0; r < 3; ++r) Vs[(4 * r + kk) * LD + j] = H[r];
      }
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) vxl[kc] = __shfl(vx, 4 * kc + kk);  // V_x[r] lives in lanes with j == r
      __syncthreads();
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) va[kc] = (j < 12) ? Vs[j * LD + 4 * kc + kk] : 0.0;
    }
    m[0] = (double)m_s0; m[1] = (double)m_s1; m[2] = (double)m_s2;
    cx[0] = (double)cx_s0; cx[1] = (double)cx_s1; cx[2] = (double)cx_s2;
    gcj = (double)g_s;
    QKEEP(va[0]); QKEEP(vxl[2]);
    QSTAMP(7);  // V_xx MFMA, gain stores, hand-off
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
constexpr int BW2_REC = 128;                   // doubles reserved for a record (symmetric layouts: stride <= 128)
constexpr int BW2_BUF = BW2_REC + CTAB_SIZE;   // one ring slot: record, then the constant operand table
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

// ---------------------------------------------------------------------------------------------
// k_backward4: k_backward2 with ONE gradient wavefront and ONE loader wavefront for FOUR trajectories
// (block = 384: matrix waves M0..M3, gradient wave G, loader wave L).  With a gradient wave per trajectory, 1024
// trajectories are 2048 wavefronts on 1024 SIMDs and every matrix wave shares its SIMD (the kernel takes 84 us
// against 64 us for 512 trajectories).  G gives a row of 16 lanes to each trajectory: lane (g, j) holds column j
// of M = [J_x | J_u] and V_x[j]; the products M^T V_x take the 12 entries of V_x by DPP row broadcasts (no
// shuffles, no butterflies), Q_u is broadcast the same way, and V_x = Q_x + K^T Q_u lands in the lane that owns
// it.  L streams the knot records of the block's four trajectories into their LDS rings (record i-3 requested in
// interval i, written in interval i-1): the matrix waves are left with the recursion and their gain stores (their
// own record loads shared the in-order memory counter with those stores: 78.7 -> 73.7 us).  Everything else as
// k_backward2.
// ---------------------------------------------------------------------------------------------
// acc += m * (vx of lane R of the caller's row of 16): one v_fmac_f64_dpp (the compiler keeps broadcast and
// multiply-add apart).  vx must have been written at least two instructions earlier (DPP read hazard): it is
// the previous knot's result here.
template <int R>
__device__ __forceinline__ double bw4_dot_step(double acc, double m, double vx) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vx), "v"(m), "n"(R));
  return acc;
}
// ---- the three roles of the backward pass over FOUR trajectories per block (k_backward4 and the persistent k_solve4).
// LDS: ring[trajectory][slot] = knot record followed by the constant operand table; kf[trajectory][parity] = K and the
// LDL^T factors handed from a matrix wave to the gradient wave.  Every role executes exactly 1 + n block barriers.
template <typename S>
__device__ __forceinline__ void bw4_fill_ctab(double (&ring)[4][4][BW2_BUF], const void *ctab, int nthreads) {
  // constant operand table behind every ring slot
  for (int t = threadIdx.x; t < CTAB_SIZE; t += nthreads) {
    const double v = (double)((const S *)ctab)[t];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ring[g][0][BW2_REC + t] = v;
      ring[g][1][BW2_REC + t] = v;
      ring[g][2][BW2_REC + t] = v;
      ring[g][3][BW2_REC + t] = v;
    }
  }
}
// G: gradients of four trajectories, one row of 16 lanes each (lane = 16 g + j).  gains / dump4: the gains of the row's
// trajectory (tiled) and its four-element dump slot; grun: the row's trajectory is being solved.  Returns Q_u^T k summed
// over the knots (every lane of the row holds it).
template <typename S>
__device__ __forceinline__ double bw4_gradient_wave(double (&ring)[4][4][BW2_BUF], double (&kf)[4][2][80], const RecLayout &L,
                                                    S *gains, S *dump4, bool grun, int n, int lane) {
  typedef typename GA<S>::v2 sv2;
  typedef typename GA<S>::ptr2 gptr2;
  const int g = lane >> 4, j = lane & 15;
  // operand addresses of lane (g, j) in ring slot 0: column j of M (12 rows) and entry j of [C_x ; C_u]
  // (the slot is a compile-time constant in gradient_step, so it folds into the ds_read offset field)
  const double *mp[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const int src = m_source_tab(r, j);
    mp[r] = &ring[g][0][(src >= 0) ? src : BW2_REC + (-1 - src)];
  }
  const double *gp = &ring[g][0][L.off_g + j];
  const bool kowner = grun && (j == 0);
  gptr2 kdst0 = (gptr2)(kowner ? gains + knot_elem<true>(n - 1, 0, 52) : dump4);
  gptr2 kdst1 = (gptr2)(kowner ? gains + knot_elem<true>(n - 1, 2, 52) : dump4 + 2);
  const long kst = kowner ? (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2 : 0;
  double vx = 0.0;  // V_x[j] (lanes j < 12)
  double QuTk = 0.0;
  __syncthreads();
  auto gradient_slot = [&](int q, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const double *f = kf[g][q & 1];
    // every LDS read of the step first, in the order of use (LDS returns in order): one exposed round trip
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) m[r] = mp[r][SLOT * BW2_BUF];
    const double gcj = gp[SLOT * BW2_BUF];
    asm volatile("" ::: "memory");
    // (K row-major from the sixteen lanes kk == 0 only -- 128 contiguous bytes per row, no bank conflicts, a quarter of the
    // bytes -- measures the same: 70.6 against 70.3 us per launch, profiles/r03_ab_backward.txt)
    const double c0 = f[4 * j], c1 = f[4 * j + 1], c2 = f[4 * j + 2], c3 = f[4 * j + 3];  // K[:, j]
    const double l10 = f[64], l20 = f[65], l30 = f[66], l21 = f[67], l31 = f[68], l32 = f[69], i0 = f[70],
                 i1 = f[71], i2 = f[72], i3 = f[73];
    asm volatile("" ::: "memory");
    // [Q_x ; Q_u][j] = [C_x ; C_u][j] + sum_r M[r][j] V_x[r], three partial sums of four rows.  (Not the
    // summation order of k_backward / k_backward2 -- rows kk, 4 + kk, 8 + kk chained, then a butterfly: that
    // order was tried here for bit-identical results across batch sizes, costs 3% and still differs in the
    // last bit elsewhere.  Results agree to ~1e-15 relative; the tests state it.)
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;
    p0 = bw4_dot_step<0>(p0, m[0], vx); p1 = bw4_dot_step<4>(p1, m[4], vx); p2 = bw4_dot_step<8>(p2, m[8], vx);
    p0 = bw4_dot_step<1>(p0, m[1], vx); p1 = bw4_dot_step<5>(p1, m[5], vx); p2 = bw4_dot_step<9>(p2, m[9], vx);
    p0 = bw4_dot_step<2>(p0, m[2], vx); p1 = bw4_dot_step<6>(p1, m[6], vx); p2 = bw4_dot_step<10>(p2, m[10], vx);
    p0 = bw4_dot_step<3>(p0, m[3], vx); p1 = bw4_dot_step<7>(p1, m[7], vx); p2 = bw4_dot_step<11>(p2, m[11], vx);
    const double ghat = gcj + ((p0 + p1) + p2);
    const double Qu0 = row_bcast<12>(ghat), Qu1 = row_bcast<13>(ghat), Qu2 = row_bcast<14>(ghat),
                 Qu3 = row_bcast<15>(ghat);
    vx = ghat + ((c0 * Qu0 + c1 * Qu1) + (c2 * Qu2 + c3 * Qu3));  // V_x = Q_x + K^T Q_u: the recurrence ends here
    double kff[4];
    ldlt4_solve_neg(Ldlt4{l10, l20, l30, l21, l31, l32, i0, i1, i2, i3}, Qu0, Qu1, Qu2, Qu3, kff);
    const double k0 = kff[0], k1 = kff[1], k2 = kff[2], k3 = kff[3];  // feed-forward (ilqr.hh:128)
    const sv2 w0 = {(S)k0, (S)k1}, w1 = {(S)k2, (S)k3};
    *kdst0 = w0;
    *kdst1 = w1;
    kdst0 -= kst;
    kdst1 -= kst;
    QuTk += Qu0 * k0 + Qu1 * k1 + Qu2 * k2 + Qu3 * k3;
  };
  auto gradient_step = [&](int q) {
    switch (q & 3) {
      case 0: gradient_slot(q, std::integral_constant<int, 0>()); break;
      case 1: gradient_slot(q, std::integral_constant<int, 1>()); break;
      case 2: gradient_slot(q, std::integral_constant<int, 2>()); break;
      default: gradient_slot(q, std::integral_constant<int, 3>()); break;
    }
  };
  // interval i: issue the loads of record i-3 (set B), gradient step of knot i+1, record i-2 (set A, loaded
  // one interval ago) into the ring; the two sets swap roles every interval
  for (int i = n - 1; i >= 0; --i) {
    if (i + 1 <= n - 1) gradient_step(i + 1);
    __syncthreads();
  }
  gradient_step(0);
  return QuTk;
}
// L: streams the knot records of the block's four trajectories (rec0..rec3: their record bases, TILED placement) into the
// rings.  A record is stride / 2 entry pairs TILE2 elements apart, and with tiles of four the block's trajectories are the
// four slots of one tile: lane l takes trajectory g = l & 3 and entry pairs l / 4 + 16 j (j = 0..3), so that one load
// instruction covers sixteen pairs of all four trajectories -- a contiguous kilobyte when they use the same record buffer
// (each trajectory has its own current buffer, hence a base per lane) -- and the lane writes its sixteen bytes to ring
// entries 2 pair, 2 pair + 1 of ring g.  Pairs beyond the record are clamped to its last pair and land in entries nobody
// reads.
//
// FREE (the fused form's product path): no block barrier inside the knot loop.  The five wavefronts of a block meet through
// 24 words of LDS instead (prog[]):
//   prog[w], w = 0..3   MG_w holds the operands of this many records in registers or is done with them (1 after its prologue,
//                       k + 2 after the knot of record k): the slots of those records may be overwritten
//   prog[4]             records the loader has placed (diagnostic)
//   prog[5]             somebody's bounded wait ran out: the block's results are void, the host is told (BatchState::host_error)
//   prog[8 + 4 g + slot] tag of ring g's slot: the ordinal t of the record it holds (record t is knot n - 1 - t), -1 before
// L writes a record's pairs, then the four tags (LDS operations of one wavefront execute in order); MG_w reads the tag of the
// slot it is about to take its next operands from and only then the operands.  L overwrites a slot once every live MG wave
// has finished the knot that read it.  With the barrier, every wavefront of the block waited for the slowest at every knot
// (1 live wave: 66.9 us per launch at N = 100, 4 live: 73.8); without it each matrix wave runs at its own pace: 68.1 with
// four live, 78.6 against 83.7 at B = 1024 (profiles/r03_ab_backward.txt).  All waits are bounded spins.
constexpr int BW4_SPIN_MAX = 1 << 22;
#ifdef QILQR_DIAG
// diagnostics build: the record ordinal whose tags the loader withholds (-1: none), so that the matrix wavefronts' bounded
// waits run out (tests/test_gpu_robustness.py)
__device__ int g_bw4_stall_rec = -1;
#endif
__device__ __forceinline__ int bw4_prog_read(int *prog, int k) { return __hip_atomic_load(&prog[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void bw4_prog_post(int *prog, int k, int v, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(&prog[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename S, bool FREE = false>
__device__ __forceinline__ void bw4_loader_wave(double (&ring)[4][4][BW2_BUF], const RecLayout &L, const S *rec0, const S *rec1,
                                                const S *rec2, const S *rec3, int n, int lane, int *prog = nullptr, int live = 0) {
  static_assert(BW2_BUF % 2 == 0 && BW2_REC == 128, "ring entries are written in aligned pairs, 64 of them per record slot");
  typedef typename GA<S>::v2 rv2;
  typedef typename GA<S>::cptr2 rptr2;
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const int g = lane & 3, npairs = L.stride / 2;
  const S *lp = (g & 2) ? ((g & 1) ? rec3 : rec2) : ((g & 1) ? rec1 : rec0);
  int poff[4];   // element offset of the lane's j-th pair inside a knot
  double *dst[4];  // its ring entries in slot 0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pu = (lane >> 2) + 16 * j, pc = pu < npairs ? pu : npairs - 1;
    poff[j] = (int)rec_elem(L, 0, 2 * pc);
    dst[j] = &ring[g][0][2 * pu];
  }
  const long knot_step = rec_elem(L, 1, 0);
  auto rec_pair = [&](int j, int i) -> rv2 { return *(rptr2)(lp + (long)i * knot_step + poff[j]); };
  auto put = [&](int j, int i, rv2 v) {
    const dv2 d = {(double)v.x, (double)v.y};
    *reinterpret_cast<dv2 *>(dst[j] + (i & 3) * BW2_BUF) = d;
  };
  rv2 q[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // (trajectories with nothing to do are streamed too: no branches in this wave)
    const rv2 a = rec_pair(j, n - 1);
    rv2 b_ = {0, 0};
    if (n >= 2) b_ = rec_pair(j, n - 2);
    if (n >= 3) q[j] = rec_pair(j, n - 3);
    put(j, n - 1, a);
    if (n >= 2) put(j, n - 2, b_);
  }
  if constexpr (FREE) {
    if (lane == 0) prog[4] = n >= 2 ? 2 : 1;
  }
  __syncthreads();  // rings and constant tables are filled
  if constexpr (FREE) {
    auto wait_slot = [&](int t) -> bool {
      if (t < 4) return true;
      int spins = 0;
      for (;;) {
        int lo = 1 << 30;
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if ((live >> w) & 1) {
            const int c = bw4_prog_read(prog, w);
            lo = c < lo ? c : lo;
          }
        if (lo >= t - 3) break;
        if (bw4_prog_read(prog, 5) || ++spins > BW4_SPIN_MAX) {
          bw4_prog_post(prog, 5, 1, lane);
          return false;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
      return true;
    };
    for (int t = 2; t < n; ++t) {
      const int i = n - 1 - t;
      if (!wait_slot(t)) return;
#pragma unroll
      for (int j = 0; j < 4; ++j) put(j, i, q[j]);
      asm volatile("" ::: "memory");
#ifdef QILQR_DIAG
      if (t != g_bw4_stall_rec)  // fault injection (qilqr_debug_set_backward_stall)
#endif
      if (lane < 4) __hip_atomic_store(&prog[8 + 4 * lane + (i & 3)], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (t + 1 < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = rec_pair(j, i - 1);
      }
      bw4_prog_post(prog, 4, t + 1, lane);
    }
    return;
  }
  for (int i = n - 1; i >= 0; --i) {
    // first the four pieces requested one interval ago, then the next four requests: the wait in front of
    // the LDS writes is for loads that are all older than anything in flight
    if (i - 2 >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) put(j, i - 2, q[j]);
    }
    if (i - 3 >= 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j] = rec_pair(j, i - 3);
      }
    }
    __syncthreads();
  }
}
// M_w: the matrix recursion of one trajectory (ring / kf row w).  cuu: the lane's entry of C_uu = 2 R (+ mu on the diagonal,
// lm_restart) in accumulator register 3 (row 12 + kk, column j >= 12), zero elsewhere.
template <typename S, bool UNROLL = false>
__device__ __forceinline__ void bw4_matrix_wave(double (&ring)[4][4][BW2_BUF], double (&kf)[4][2][80], const RecLayout &L, int w,
                                                bool run, S *gains, S *dump4, double cuu, int n, int lane,
                                                unsigned long long *stamps_out) {
  typedef typename GA<S>::v2 sv2;
  typedef typename GA<S>::ptr2 gptr2;
  // the matrix waves are the block's critical chain: issue them ahead of the gradient / loader wave (and of other blocks'
  // helper waves) on their SIMD.  Nothing at one block per CU (83.1 us either way), +1.5 % of a solve at four blocks per CU
  __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, kk = lane >> 4;
  int off[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int src;
    if (k < 3) src = m_source_tab(4 * k + kk, j);
    else src = (j < 12) ? cxx_source_tab(L, 4 * (k - 3) + kk, j) : -1 - CTAB_ZERO;
    off[k] = (src >= 0) ? src : BW2_REC + (-1 - src);
  }
  const bool gowner = run && (kk == 0 && j < 12);
  const int ge0 = 4 + 4 * j;
  gptr2 gdst0 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0, 52) : dump4);
  gptr2 gdst1 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0 + 2, 52) : dump4 + 2);
  const long gstep = gowner ? (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2 : 0;
  double va[3] = {0.0, 0.0, 0.0};  // V_xx[j][4 kc + kk]  (A operand)
  __syncthreads();  // rings and constant tables are filled
  if (!run) {
    // this trajectory has nothing to do in this round: keep the block's barriers company
    for (int i = n - 1; i >= 0; --i) __syncthreads();
    return;
  }
  double m[3], cx[3];
  {
    const double *buf = ring[w][(n - 1) & 3];
    m[0] = buf[off[0]]; m[1] = buf[off[1]]; m[2] = buf[off[2]];
    cx[0] = buf[off[3]]; cx[1] = buf[off[4]]; cx[2] = buf[off[5]];
  }
  // The knot loop is sensitive to where its instruction stream sits: the same code shifted by 4 bytes (mod 8) runs 7 %
  // slower (71.5 -> 77 us per launch at B = 1024; MI355X_MICROARCH.md, "code-placement sensitivity").  Pin it to a
  // 64-byte boundary.
  asm volatile(".p2align 6");  // (the loop is sensitive to where it sits; phase 0 behind a 64-byte boundary measured fastest of 0..7 in round 3: profiles/r03_ab_backward.txt)
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev, real0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  // One knot.  (mc, cc): its operands, in registers; (mn, cn): the operands of the next knot (i - 1), requested here from ring
  // slot `ns` (filled during the previous interval); par: i & 1, the hand-off buffer.  ns and par are ints in the rolled loop
  // and compile-time constants in the unrolled one (immediate offsets of the LDS instructions).
  const double *rp[6];  // the lane's six operand addresses in ring slot 0
#pragma unroll
  for (int k = 0; k < 6; ++k) rp[k] = &ring[w][0][off[k]];
  double *const kfw = &kf[w][0][0];
  auto knot = [&](auto ns, auto par, double (&mc)[3], double (&cc)[3], double (&mn)[3], double (&cn)[3]) {
    const int so = (int)ns * BW2_BUF;
    mn[0] = rp[0][so]; mn[1] = rp[1][so]; mn[2] = rp[2][so];
    cn[0] = rp[3][so]; cn[1] = rp[4][so]; cn[2] = rp[5][so];
    const d4 T = bw_tile_T(va, mc);
    QKEEP(T[0]); QKEEP(T[3]);
    QSTAMP(0);  // ring reads issued, T = V M
    d4 H = bw_tile_H(mc, T, cc, cuu);
    QKEEP(H[0]); QKEEP(H[3]);
    QSTAMP(1);  // H = C + M^T T
    double Quu[16], Qu_unused[4], col[4];
    gather_rows(H[3], col);
    bcast_quu_row<0>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<1>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<2>(col, 0.0, Quu, Qu_unused);
    bcast_quu_row<3>(col, 0.0, Quu, Qu_unused);
    QKEEP(Quu[0]); QKEEP(Quu[15]); QKEEP(col[3]);
    QSTAMP(2);  // gather + Q_uu broadcast
    const Ldlt4 f4 = ldlt4_factor(Quu);  // (ilqr.hh:126; unpivoted: see ldlt4_factor)
    double kcol[4];
    ldlt4_solve_neg(f4, col[0], col[1], col[2], col[3], kcol);  // K[:, j] (ilqr.hh:127)
    QKEEP(kcol[0]); QKEEP(kcol[3]);
    QSTAMP(4);  // LDL^T + solve
    {
      const sv2 w0 = {(S)kcol[0], (S)kcol[1]}, w1 = {(S)kcol[2], (S)kcol[3]};
      *gdst0 = w0;
      *gdst1 = w1;
      gdst0 -= gstep;
      gdst1 -= gstep;
    }
    // hand K and the factors to G (the four lanes of a column hold the same K[:, j]: same address, same data)
    double *f = kfw + (int)par * 80;
    f[4 * j] = kcol[0]; f[4 * j + 1] = kcol[1]; f[4 * j + 2] = kcol[2]; f[4 * j + 3] = kcol[3];
    if (lane == 0) {
      f[64] = f4.l10; f[65] = f4.l20; f[66] = f4.l30; f[67] = f4.l21; f[68] = f4.l31; f[69] = f4.l32;
      f[70] = f4.i0; f[71] = f4.i1; f[72] = f4.i2; f[73] = f4.i3;
    }
    QSTAMP(5);  // gain stores, hand-off to G
    // V_xx = Q_xx + Q_xu K: A[j][kk] = Q_xu[j][kk] = H[12 + kk][j] is accumulator register 3
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(H[3], sel4(kcol, kk), H, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    QKEEP(va[0]); QKEEP(mn[2]);
    QSTAMP(6);  // V_xx MFMA, next operands
    __syncthreads();
    QSTAMP(7);  // barrier
  };
  // Two forms of the loop.  Rolled: 147 instructions per knot.  Unrolled by four -- ring slot and hand-off parity as immediate
  // offsets, the two operand register sets alternating: no copies, no address arithmetic, 124 instructions per knot.  Which
  // is faster depends on what bounds the wave (profiles/r03_ab_backward.txt).  With one block per CU (B = 1024) a matrix wave
  // is alone on its SIMD and bound by the LATENCIES between its instructions -- seven dependent matrix instructions, the
  // reciprocal chains of the factorisation, the cross-lane gathers --, 23 fewer instructions return nothing and the four
  // times longer loop body costs instruction fetch: 71.4 us per launch unrolled against 70.2 rolled, every code phase tried.
  // With four blocks per CU (B > 4096) the SIMD interleaves four matrix waves and is bound by what they ISSUE: there the
  // unrolled loop is the faster one.  The instantiation decides (UNROLL = the many-blocks build of k_backward4).
  double mn[3], cn[3];
  int i = n - 1;
  if constexpr (UNROLL) {
    // the first n mod 4 knots, until the knot index is 3 mod 4
    for (; i >= 0 && (i & 3) != 3; --i) {
      knot((i > 0 ? i - 1 : 0) & 3, i & 1, m, cx, mn, cn);
#pragma unroll
      for (int k = 0; k < 3; ++k) { m[k] = mn[k]; cx[k] = cn[k]; }
    }
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    typedef std::integral_constant<int, 2> C2;
    typedef std::integral_constant<int, 3> C3;
    for (; i >= 3; i -= 4) {
      knot(C2(), C1(), m, cx, mn, cn);    // knot 4 q + 3 (slot 3); next operands from slot 2
      knot(C1(), C0(), mn, cn, m, cx);    // knot 4 q + 2
      knot(C0(), C1(), m, cx, mn, cn);    // knot 4 q + 1
      knot(C3(), C0(), mn, cn, m, cx);    // knot 4 q; the next pass starts in slot 3 (after knot 0: read and never used)
    }
  } else {
    // (the rolled loop is written out, not built from `knot`: the same statements through the lambda schedule 1.3 us per
    // launch slower -- this loop is that sensitive to the order the compiler picks)
    for (; i >= 0; --i) {
      // operands of knot i-1, for the next iteration (the slot was filled during the previous interval)
      const double *nb = ring[w][(i > 0 ? i - 1 : 0) & 3];
      const double m_n0 = nb[off[0]], m_n1 = nb[off[1]], m_n2 = nb[off[2]], cx_n0 = nb[off[3]], cx_n1 = nb[off[4]],
                   cx_n2 = nb[off[5]];
      const d4 T = bw_tile_T(va, m);
      QKEEP(T[0]); QKEEP(T[3]);
      QSTAMP(0);  // ring reads issued, T = V M
      d4 H = bw_tile_H(m, T, cx, cuu);
      QKEEP(H[0]); QKEEP(H[3]);
      QSTAMP(1);  // H = C + M^T T
      double Quu[16], Qu_unused[4], col[4];
      gather_rows(H[3], col);
      bcast_quu_row<0>(col, 0.0, Quu, Qu_unused);
      bcast_quu_row<1>(col, 0.0, Quu, Qu_unused);
      bcast_quu_row<2>(col, 0.0, Quu, Qu_unused);
      bcast_quu_row<3>(col, 0.0, Quu, Qu_unused);
      QKEEP(Quu[0]); QKEEP(Quu[15]); QKEEP(col[3]);
      QSTAMP(2);  // gather + Q_uu broadcast
      const Ldlt4 f4 = ldlt4_factor(Quu);  // (ilqr.hh:126; unpivoted: see ldlt4_factor)
      double kcol[4];
      ldlt4_solve_neg(f4, col[0], col[1], col[2], col[3], kcol);  // K[:, j] (ilqr.hh:127)
      QKEEP(kcol[0]); QKEEP(kcol[3]);
      QSTAMP(4);  // LDL^T + solve
      {
        const sv2 w0 = {(S)kcol[0], (S)kcol[1]}, w1 = {(S)kcol[2], (S)kcol[3]};
        *gdst0 = w0;
        *gdst1 = w1;
        gdst0 -= gstep;
        gdst1 -= gstep;
      }
      // hand K and the factors to G (the four lanes of a column hold the same K[:, j]: same address, same data)
      double *f = kf[w][i & 1];
      f[4 * j] = kcol[0]; f[4 * j + 1] = kcol[1]; f[4 * j + 2] = kcol[2]; f[4 * j + 3] = kcol[3];
      if (lane == 0) {
        f[64] = f4.l10; f[65] = f4.l20; f[66] = f4.l30; f[67] = f4.l21; f[68] = f4.l31; f[69] = f4.l32;
        f[70] = f4.i0; f[71] = f4.i1; f[72] = f4.i2; f[73] = f4.i3;
      }
      QSTAMP(5);  // gain stores, hand-off to G
      // V_xx = Q_xx + Q_xu K: A[j][kk] = Q_xu[j][kk] = H[12 + kk][j] is accumulator register 3
      H = __builtin_amdgcn_mfma_f64_16x16x4f64(H[3], sel4(kcol, kk), H, 0, 0, 0);
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
      m[0] = m_n0; m[1] = m_n1; m[2] = m_n2;
      cx[0] = cx_n0; cx[1] = cx_n1; cx[2] = cx_n2;
      QKEEP(va[0]); QKEEP(m[2]);
      QSTAMP(6);  // V_xx MFMA, next operands
      __syncthreads();
      QSTAMP(7);  // barrier
    }
  }
#ifdef QILQR_STAMPS
  {
    // slot 3 (no section of wave M uses it): the loop's duration on the constant 100 MHz clock, so that
    // cycles / time gives the shader clock the loop ran at
    unsigned long long real1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");
    stamp_sum[3] = (real1 - real0) & 0xfffffull;
  }
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
}

// MG_w: matrix AND gradient recursion of one trajectory in one wavefront (k_backward4<.., FUSED = true>): the arithmetic of the
// one-wavefront kernel k_backward<true> -- gradient by three multiply-adds and two permlane butterflies, k solved in lane 12
// with the lane's own factors, V_x = Q_x + K^T Q_u -- with its seven operands from the LDS ring the loader wave fills
// (ring row w; no gradient wavefront, no hand-off of K and the factors; no block barrier in the knot loop: the loader tags
// every ring slot it fills, the wave looks at the tag of its next record's slot before it reads the operands, every wait a
// bounded spin).  Returns Q_u^T k summed over the knots.
// The knot is SOFTWARE-PIPELINED around the six matrix instructions (round 4).  A lone wavefront issues in
// order, and what profiles/r04_knot_anatomy.txt shows is a knot whose pieces simply add up (1640 cycles: 6 + 1 MFMA 500, the
// 4x4 solve 300, tag check and operand reads 280, gather and broadcasts 140, stores / Q_u^T k / V_x / shuffles 140, ...): the
// compiler issues T's three products back to back, then everything else.  But a chained v_mfma_f64_16x16x4_f64 cannot issue
// before its predecessor has finished (64 cycles, mfma_chain.hip), and in between the wavefront is free to issue anything
// that does not touch the tile -- so everything that is NOT on the chain V_xx -> T -> H -> gather -> solve -> V_xx is issued
// in those gaps, one group behind each product, the groups held in place by scheduling barriers:
//     T1 | V_x of the PREVIOUS knot (its K and Q_u are carried over)     T2 | its shuffles, Q_u^T k
//     T3 | the previous knot's gain stores, H's start values             H1 | M^T V_x, three multiply-adds
//     H2 | the two butterflies, Q_x / Q_u                                H3 | tag check, next operands from the ring, progress
// then the chain's own part: row gather, Q_uu broadcasts, LDL^T, solve, operand select, V_xx.  The first knot carries zeros in
// (V_x = 0, a store to the dump slot); the last knot's tail runs behind the loop.  Same arithmetic, same order of operations
// per value as the round-3 loop.  (Measured, profiles/microbench/mfma_shadow.hip: between two chained products a wavefront's own
// integer / move / DPP / permlane instructions cost 1.5-3 cycles each instead of 4-5; fp64 instructions hide nothing -- they
// share the double-precision units with the products.)
template <typename S>
__device__ __forceinline__ double bw4_fused_wave(double (&ring)[4][4][BW2_BUF], const RecLayout &L, int w, bool run, S *gains, S *dump4,
                                                          double cuu, int n, int lane, int *prog, unsigned long long *stamps_out = nullptr) {
  typedef typename GA<S>::v2 sv2;
  typedef typename GA<S>::ptr2 gptr2;
  __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, kk = lane >> 4;
  int off[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int src;
    if (k < 3) src = m_source_tab(4 * k + kk, j);
    else if (k < 6) src = (j < 12) ? cxx_source_tab(L, 4 * (k - 3) + kk, j) : -1 - CTAB_ZERO;
    else src = L.off_g + j;
    off[k] = (src >= 0) ? src : BW2_REC + (-1 - src);
  }
  const bool gowner = run && (kk == 0 && j <= 12);
  const int ge0 = (j < 12) ? 4 + 4 * j : 0;
  // the gains of a knot are stored one iteration late: st* = where the carried gains go (the dump slot in front of the first knot)
  gptr2 st0 = (gptr2)dump4, st1 = (gptr2)(dump4 + 2);
  gptr2 nx0 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0, 52) : dump4);
  gptr2 nx1 = (gptr2)(gowner ? gains + knot_elem<true>(n - 1, ge0 + 2, 52) : dump4 + 2);
  const long gstep = gowner ? (knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2 : 0;
  double va[3] = {0.0, 0.0, 0.0};   // V_xx[j][4 kc + kk]  (A operand)
  double vxl[3] = {0.0, 0.0, 0.0};  // V_x[4 kc + kk]
  double QuTk = 0.0;
  __syncthreads();  // rings and constant tables are filled
  if (!run) return 0.0;
  double m[3], cx[3], gcj;
  {
    const double *buf = ring[w][(n - 1) & 3];
    m[0] = buf[off[0]]; m[1] = buf[off[1]]; m[2] = buf[off[2]];
    cx[0] = buf[off[3]]; cx[1] = buf[off[4]]; cx[2] = buf[off[5]];
    gcj = buf[off[6]];
  }
  bw4_prog_post(prog, w, 1, lane);  // record 0 is in registers (the loader may reuse its slot)
  double kp[4] = {0.0, 0.0, 0.0, 0.0}, Qup[4] = {0.0, 0.0, 0.0, 0.0}, ghp = 0.0;  // the previous knot's K column, Q_u, Q_x
#define QSB() __builtin_amdgcn_sched_barrier(0)
  asm volatile(".p2align 6");
  bool dead = false;
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  for (int i = n - 1; i >= 0; --i) {
    const double *nb = ring[w][(i > 0 ? i - 1 : 0) & 3];
    const int slot_word = 8 + 4 * w + ((i > 0 ? i - 1 : 0) & 3), want = n - 1 - i + 1;
    const unsigned tag_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int *)&prog[slot_word];
    int tag = bw4_prog_read(prog, slot_word);  // an ordinary load: the compiler keeps count of it
    QSB();
    d4 T = {0.0, 0.0, 0.0, 0.0};
    d4 H;
    double Quu[16], Qu[4], col[4], rhs[4], ghat, h3;
    double m_n0, m_n1, m_n2, cx_n0, cx_n1, cx_n2, g_n;
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[0], m[0], T, 0, 0, 0);
    QSB();
    // K^T Q_u of the previous knot: V_x = Q_x + K^T Q_u in every lane, and in lane 12 -- whose column is k and whose right-hand
    // side was Q_u -- the same sum is Q_u^T k (one sum for both; nobody reads the other lanes' Q_u^T k)
    const double ktq = kp[0] * Qup[0] + kp[1] * Qup[1] + kp[2] * Qup[2] + kp[3] * Qup[3];
    const double vx = ghp + ktq;
    QSB();
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[1], m[1], T, 0, 0, 0);
    QSB();
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) vxl[kc] = __shfl(vx, 4 * kc + kk);  // V_x[r] lives in lanes with j == r
    QuTk += ktq;  // (lane 12's is Q_u^T k)
    QSB();
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(va[2], m[2], T, 0, 0, 0);
    QSB();
    {
      const sv2 w0 = {(S)kp[0], (S)kp[1]}, w1 = {(S)kp[2], (S)kp[3]};
      *st0 = w0;
      *st1 = w1;
      st0 = nx0; st1 = nx1;
      nx0 -= gstep; nx1 -= gstep;
    }
    H = d4{cx[0], cx[1], cx[2], cuu};
    QKEEP(T[0]); QKEEP(vxl[0]); QKEEP(vxl[2]); QKEEP(QuTk);
    QSTAMP(0);  // T (3 MFMA) with the previous knot's V_x, shuffles, Q_u^T k, stores in the gaps
    QSB();
    // (the kc = 2 product first: rows 12..15 of H -- result register 3 -- receive nothing from the other two, J_u being zero in
    // rows 0..7, so the register is final one product early; the order is part of the arithmetic: every build adds in it)
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[2], T[2], H, 0, 0, 0);
    QSB();
    double part = m[0] * vxl[0] + m[1] * vxl[1] + m[2] * vxl[2];  // [Q_x ; Q_u] = [C_x ; C_u] + M^T V_x
    QSB();
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[0], T[0], H, 0, 0, 0);
    QSB();
    part = xor16_sum(part);
    part = xor32_sum(part);
    ghat = gcj + part;
    QSB();
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(m[1], T[1], H, 0, 0, 0);
    QKEEP(H[3]); QKEEP(ghat);
    QSTAMP(1);  // H (3 MFMA) with M^T V_x and the butterflies in the gaps
    QSB();
    {
      int tag_s = __builtin_amdgcn_readfirstlane(tag);
      if (__builtin_expect(i > 0 && want >= 2 && !dead && tag_s != want, 0)) {
        int spins = 0;
        do {
          asm volatile("ds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0" : "=&v"(tag), "=s"(tag_s) : "v"(tag_addr) : "memory");
          if (++spins > BW4_SPIN_MAX) dead = true;
        } while (tag_s != want && !dead);
      }
      asm volatile("" ::: "memory");
      m_n0 = nb[off[0]]; m_n1 = nb[off[1]]; m_n2 = nb[off[2]]; cx_n0 = nb[off[3]]; cx_n1 = nb[off[4]]; cx_n2 = nb[off[5]]; g_n = nb[off[6]];
      bw4_prog_post(prog, w, n - 1 - i + 2, lane);  // (the LDS executes a wavefront's operations in order: behind the reads)
    }
    QKEEP(m_n0); QKEEP(g_n);
    QSTAMP(2);  // tag check, next operands, progress
    QSB();
    gather_rows(H[3], col);
    h3 = H[3];
    bcast_quu_row<0>(col, ghat, Quu, Qu);
    bcast_quu_row<1>(col, ghat, Quu, Qu);
    bcast_quu_row<2>(col, ghat, Quu, Qu);
    bcast_quu_row<3>(col, ghat, Quu, Qu);
#pragma unroll
    for (int a = 0; a < 4; ++a) rhs[a] = (j == 12) ? Qu[a] : col[a];  // lane 12: feed-forward
    QKEEP(rhs[0]); QKEEP(rhs[3]); QKEEP(Quu[15]);
    QSTAMP(3);  // row gather, Q_uu / Q_u broadcasts, right-hand sides
    const Ldlt4 f4 = ldlt4_factor(Quu);
    double kcol[4];
    ldlt4_solve_neg(f4, rhs[0], rhs[1], rhs[2], rhs[3], kcol);  // K[:, j] (ilqr.hh:127); k in lane 12 (:128)
    QKEEP(kcol[0]); QKEEP(kcol[3]);
    QSTAMP(4);  // LDL^T and solve
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(h3, sel4(kcol, kk), H, 0, 0, 0);  // V_xx = Q_xx + Q_xu K (A = rows 12..15 of H)
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    QKEEP(va[0]); QKEEP(va[2]);
    QSTAMP(5);  // operand select, V_xx MFMA
    m[0] = m_n0; m[1] = m_n1; m[2] = m_n2;
    cx[0] = cx_n0; cx[1] = cx_n1; cx[2] = cx_n2;
    gcj = g_n;
#pragma unroll
    for (int a = 0; a < 4; ++a) { kp[a] = kcol[a]; Qup[a] = Qu[a]; }
    ghp = ghat;
  }
#undef QSB
  {  // the last knot's tail
    const sv2 w0 = {(S)kp[0], (S)kp[1]}, w1 = {(S)kp[2], (S)kp[3]};
    *st0 = w0;
    *st1 = w1;
    QuTk += kp[0] * Qup[0] + kp[1] * Qup[1] + kp[2] * Qup[2] + kp[3] * Qup[3];
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
  if (dead) bw4_prog_post(prog, 5, 1, lane);
  return bcast_lane(QuTk, 12);
}

// WAVES: register budget in waves per SIMD.  5 (90 registers, nothing spilled): three blocks per CU, the fastest single
// block; 6 (80 registers, four of them spilled outside the knot loop): four blocks per CU -- with 33 KB of LDS per block the
// registers are what decides -- for the launches that have more than three blocks per CU to run (B = 8192 in two parts:
// 404 000 -> 412 000 solves/s; nothing at 4096, -0.5 % at 1024)
// FUSED: five wavefronts per block -- MG_0..MG_3 (matrix and gradient recursion of a trajectory in one wavefront, bw4_fused_wave)
// and the loader L -- instead of six (M_0..M_3, G, L).
// the LDS of a block of k_backward4 (backward4_body.inc declares it unless the including kernel has: BW4_LDS_DECLARED).  ring: four slots
// per trajectory -- in interval i the matrix wave reads slot (i-1) & 3 and writes slot (i-2) & 3 while G reads slot (i+1) & 3
#define BW4_DECLARE_LDS                                                  \
  __shared__ int s_run[4], s_cur[4], s_iters[4], s_act[4];               \
  __shared__ double s_cost[4];                                           \
  __shared__ __attribute__((aligned(16))) double ring[4][4][BW2_BUF];    \
  __shared__ double kf[4][2][80];                                        \
  __shared__ int prog[24];
#define QILQR_CAT_(a, b) a##b
#define QILQR_CAT(a, b) QILQR_CAT_(a, b)
template <typename S, int WAVES, bool FUSED = false, bool FREE = false>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_backward4(ModelConsts<double> c, SolveParams p, BatchState st, int B, int n,
                                                   int force) {
  // (the body lives in a file of its own because k_backward_rollout contains it too, as statements of the kernel function:
  // called as a device function it loses what the compiler knows about pointers that come from kernel arguments -- every
  // global access becomes a flat one.  Round 3 saw the six-wavefront form's results change that way; the cause was a merged
  // conditional store the compiler got wrong with the workspace pointers in scratch: store_settled / arm_line_search above)
#define BW4_RETURN return
#include "backward4_body.inc"
#undef BW4_RETURN
}

// ---------------------------------------------------------------------------------------------
// k_rollout: thread b.  traj[cur] + gains + alpha -> traj[cur ^ 1]
// ---------------------------------------------------------------------------------------------
template <typename S, int INTEG>
__global__ __launch_bounds__(64) void k_rollout(ModelConsts<S> c, BatchState st, int B, int n,
                                                int need_flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (need_flag && !(st.flags[b] & need_flag)) return;
  const int cur = st.cur[b];
  rollout_problem<true, S, INTEG>(c, (const S *)st.traj[cur] + knot_base<true>(b, n, 18),
                        (const S *)st.gains + knot_base<true>(b, n, 52), (S)st.alpha[b],
                        (S *)st.traj[cur ^ 1] + knot_base<true>(b, n, 18), n);
}

// ---------------------------------------------------------------------------------------------
// k_rollout3: the rollout with THREE cooperating wavefronts per 64 trajectories (block = 192).
// A single wavefront issues one fp64 instruction per ~5-9 cycles whatever the instruction-level
// parallelism (profiles/microbench), so the serial per-knot chain is split into the two halves that
// are independent inside one knot, and the operand loads are taken off both:
//   wave Y (pose):     T_{i+1} = T_i Exp(dt v_i), then the pose part of x_{i+1} (-) xnom_{i+1}
//   wave X (control):  rho_i = Jl^-1 td_i, u_i = u_nom + alpha k + K dx_i, v_{i+1} = v_i + dt a(q_i, v_i, u_i)
//   wave L (loader):   streams the next knots' nominal point and gains (35 sixteen-byte loads per lane and
//                      knot, whose issue alone cost the control wave a third of a knot) two knots ahead
//                      through registers into a double-buffered LDS image (same [pair][lane] order as the
//                      tiled global layout: conflict-free), and the nominal pose that the pose wave needs one
//                      knot earlier into a second small image.  Waves X and Y read LDS only.
// X and Y trade 11 + 6 scalars per knot through LDS (double-buffered).
//   iteration i:  L: issue loads of knot i+2 (+ pose of knot i+3); write knot i+1 -> bx[(i+1)&1],
//                    pose of knot i+2 -> by[(i+2)&1]
//                 X: operands of knot i from bx[i&1];  Y: nominal pose of knot i+1 from by[(i+1)&1]
//   one barrier per knot.  The arithmetic is the same sequence of operations as rollout_problem (k_rollout).
// (Two re-partitions were measured in round 1 and removed from the library: the pair without the loader,
// and a four-wave form with the pose wave cut into compose and Log -- DESIGN.md section 4.)
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(192) void k_rollout3(ModelConsts<S> c, BatchState st, int B, int n, int need_flag) {
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0: X, 1: Y, 2: L
  const int b = blockIdx.x * 64 + lane;
  const bool live = (b < B) && (!need_flag || (st.flags[b < B ? b : 0] & need_flag));
  if (__ballot(live) == 0ull) return;  // identical in the three waves: block-uniform
  // The pose and the control wavefront are a serial chain that a whole sub-batch waits for, and with sub-batches on their own streams they
  // share their SIMDs with other sub-batches' backward passes, whose matrix wavefronts issue at priority 3: at the default priority this
  // kernel took 149 us per launch at B = 8192 against 81-105 with the chip to itself.
  if (role < 2) __builtin_amdgcn_s_setprio(3);
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev, real_entry, real0 = 0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real_entry)::"memory");
  auto stamp_flush = [&]() {
    if (lane == 0 && st.stamps)
      for (int k = 0; k < 8; ++k) st.stamps[((long)blockIdx.x * 3 + role) * 8 + k] = stamp_sum[k];
  };
#endif
  const int bs = (b < B) ? b : (B - 1);
  const int cur = st.cur[bs];
  const S *traj = (const S *)st.traj[cur] + knot_base<true>(bs, n, 18);
  const S *gains = (const S *)st.gains + knot_base<true>(bs, n, 52);
  S *out = (S *)st.traj[cur ^ 1] + knot_base<true>(bs, n, 18);

  typedef S sv2 __attribute__((ext_vector_type(2)));
  __shared__ sv2 bx[2][35][64];  // [parity][pair: 0..8 nominal knot, 9..34 gains][lane]
  __shared__ sv2 by[2][4][64];   // [parity][pair 0..3 of the nominal knot = time, t, q][lane]
  __shared__ S sh[2][17][64];    // X <-> Y exchange: [parity][0..3 q | 4..6 td | 7..9 th | 10 c | 11..16 v][lane]

  // Each role runs its own loop (so that the register allocator sees three disjoint live ranges);
  // all three execute exactly 1 + n barriers.
  if (role == 2) {
    // ------------------------------------------------------------------ L: loader
    const sv2 *tp = reinterpret_cast<const sv2 *>(traj);   // pair k of knot i: tp[(i * 9 + k) * TILE]
    const sv2 *gp = reinterpret_cast<const sv2 *>(gains);  //                   gp[(i * 26 + k) * TILE]
    sv2 ra[35], rb[35], pa[4], pb[4];
    // Lanes whose trajectory is not being rolled out this round request nothing after the first knots (in the
    // late rounds a tile holds a handful of live trajectories and 16-byte sectors of the others would be most
    // of the kernel's HBM traffic); their registers keep the first knots' operands, so the other two waves go
    // on computing finite values for them that nobody stores.
    auto load_ops = [&](int k, sv2 (&r)[35], bool every_lane) {
      if (k < n && (every_lane || live)) {
#pragma unroll
        for (int e = 0; e < 9; ++e) r[e] = tp[((long)k * 9 + e) * TILE];
#pragma unroll
        for (int e = 0; e < 26; ++e) r[9 + e] = gp[((long)k * 26 + e) * TILE];
      }
    };
    auto load_pose = [&](int k, sv2 (&r)[4], bool every_lane) {
      if (k < n && (every_lane || live)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = tp[((long)k * 9 + e) * TILE];
      }
    };
    load_ops(0, ra, true);
    load_pose(n > 1 ? 1 : 0, pa, true);
#pragma unroll
    for (int e = 0; e < 35; ++e) bx[0][e][lane] = ra[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) by[1][e][lane] = pa[e];
#pragma unroll
    for (int e = 0; e < 35; ++e) rb[e] = ra[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) pb[e] = pa[e];
    load_ops(1, ra, false);   // written during iteration 0
    load_pose(2, pa, false);  // written during iteration 0
    __syncthreads();
#ifdef QILQR_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    auto knot = [&](int i, sv2 (&rc)[35], sv2 (&rn)[35], sv2 (&pc)[4], sv2 (&pn)[4]) {
      load_ops(i + 2, rn, false);   // consumed by X at iteration i + 2
      load_pose(i + 3, pn, false);  // consumed by Y at iteration i + 2
      QSTAMP(0);  // L: load issue
      if (i + 1 < n) {
#pragma unroll
        for (int e = 0; e < 35; ++e) bx[(i + 1) & 1][e][lane] = rc[e];  // knot i + 1
      }
      if (i + 2 < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) by[i & 1][e][lane] = pc[e];  // nominal pose of knot i + 2
      }
      QSTAMP(1);  // L: wait for the loads of the previous iteration, LDS writes
      __syncthreads();
      QSTAMP(5);  // L: barrier
    };
    for (int i = 0; i < n; i += 2) {
      knot(i, ra, rb, pa, pb);
      if (i + 1 < n) knot(i + 1, rb, ra, pb, pa);
    }
#ifdef QILQR_STAMPS
    stamp_flush();
#endif
    return;
  }

  // the time step of a lane that is not being rolled out is zero: its state stays where it starts, next to the
  // first knots' nominal values the loader keeps giving it, on the cheap branches of Exp and Log
  const S dtl = live ? c.dt : S(0);
  S t[3], q[4], v[6], td[3] = {0, 0, 0}, th[3] = {0, 0, 0}, cj = 0;
  {
    S p0[18];
    load_knot<true>(traj, 0, 18, p0);
    t[0] = p0[1]; t[1] = p0[2]; t[2] = p0[3];
    q[0] = p0[5]; q[1] = p0[6]; q[2] = p0[7]; q[3] = p0[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = p0[8 + a];
    if (role == 1) {
      const S qn[4] = {p0[5], p0[6], p0[7], p0[4]};
      se3_rminus_part1(t, q, p0 + 1, qn, td, th, cj);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        sh[0][4 + a][lane] = td[a];
        sh[0][7 + a][lane] = th[a];
      }
      sh[0][10][lane] = cj;
      if (live) {
        const S po[8] = {0, t[0], t[1], t[2], q[3], q[0], q[1], q[2]};
#pragma unroll
        for (int e = 1; e < 8; ++e) out[knot_elem<true>(0, e, 18)] = po[e];
      }
    }
  }
  __syncthreads();

  if (role == 0) {
    // ------------------------------------------------------------------ X: control + velocity
    const S alpha = (S)st.alpha[bs];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      td[a] = sh[0][4 + a][lane];
      th[a] = sh[0][7 + a][lane];
    }
    cj = sh[0][10][lane];
#ifdef QILQR_STAMPS
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    for (int i = 0; i < n; ++i) {
      const int par = (i + 1) & 1;
      const bool more = (i + 1 < n);
      S pt[18], g[52];
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const sv2 w = bx[i & 1][e][lane];
        pt[2 * e] = w[0];
        pt[2 * e + 1] = w[1];
      }
#pragma unroll
      for (int e = 0; e < 26; ++e) {
        const sv2 w = bx[i & 1][9 + e][lane];
        g[2 * e] = w[0];
        g[2 * e + 1] = w[1];
      }
      QKEEP(pt[17]); QKEEP(g[51]); QKEEP(g[0]);
      QSTAMP(0);  // X: operands from LDS
      S dx[12];
      se3_rminus_part2(td, th, cj, dx);
      dx[3] = th[0]; dx[4] = th[1]; dx[5] = th[2];
#pragma unroll
      for (int a = 0; a < 6; ++a) dx[6 + a] = v[a] - pt[8 + a];
      QKEEP(dx[0]); QKEEP(dx[11]);
      QSTAMP(1);  // X: rho = Jl^-1 td, dx
      S u[4];
      control_law(pt, g, alpha, dx, u);
      QKEEP(u[0]); QKEEP(u[3]);
      QSTAMP(2);  // X: control law
      if (live) {
        out[knot_elem<true>(i, 0, 18)] = pt[0];
#pragma unroll
        for (int a = 0; a < 6; ++a) out[knot_elem<true>(i, 8 + a, 18)] = v[a];
#pragma unroll
        for (int a = 0; a < 4; ++a) out[knot_elem<true>(i, 14 + a, 18)] = u[a];
      }
      if (more) {
        S acc[6];
        body_acceleration_fast(c, q, v, u, acc);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          v[a] = v[a] + dtl * acc[a];
          sh[par][11 + a][lane] = v[a];
        }
      }
      QSTAMP(4);  // X: stores, acceleration, velocity update, LDS write
      __syncthreads();
      QSTAMP(5);  // X: barrier
      if (more) {
#pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = sh[par][a][lane];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          td[a] = sh[par][4 + a][lane];
          th[a] = sh[par][7 + a][lane];
        }
        cj = sh[par][10][lane];
      }
      QKEEP(cj); QKEEP(q[0]);
      QSTAMP(6);  // X: LDS read of Y's results
    }
#ifdef QILQR_STAMPS
    {
      // slot 3 (unused by X): the loop on the constant 100 MHz clock (low 20 bits) | entry -> loop (next 20)
      unsigned long long real1;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");
      stamp_sum[3] = ((real1 - real0) & 0xfffffull) | (((real0 - real_entry) & 0xfffffull) << 20);
    }
#endif
  } else {
    // ------------------------------------------------------------------ Y: pose
    RolloutSeries<S> sr;  // the series coefficients, in vector registers for the whole loop
    sr.load();
#ifdef QILQR_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    for (int i = 0; i < n; ++i) {
      const int par = (i + 1) & 1;
      const bool more = (i + 1 < n);
      if (more) {
        S pnm[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const sv2 w = by[par][e][lane];  // nominal pose of knot i + 1
          pnm[2 * e] = w[0];
          pnm[2 * e + 1] = w[1];
        }
        S tau[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) tau[a] = dtl * v[a];  // pose integrates with the OLD velocity
        QKEEP(pnm[7]);
        QSTAMP(0);  // Y: nominal pose from LDS
        se3_rplus_fast(t, q, tau, sr);
        QKEEP(t[0]); QKEEP(q[3]);
        QSTAMP(1);  // Y: T <- T Exp(dt v)
        const S qn[4] = {pnm[5], pnm[6], pnm[7], pnm[4]};
        se3_rminus_part1(t, q, pnm + 1, qn, td, th, cj, sr);
        QKEEP(td[0]); QKEEP(th[2]); QKEEP(cj);
        QSTAMP(2);  // Y: pose part of x (-) xnom
#pragma unroll
        for (int a = 0; a < 4; ++a) sh[par][a][lane] = q[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          sh[par][4 + a][lane] = td[a];
          sh[par][7 + a][lane] = th[a];
        }
        sh[par][10][lane] = cj;
        if (live) {
          const S po[8] = {0, t[0], t[1], t[2], q[3], q[0], q[1], q[2]};
#pragma unroll
          for (int e = 1; e < 8; ++e) out[knot_elem<true>(i + 1, e, 18)] = po[e];
        }
      }
      QSTAMP(4);  // Y: LDS write, stores
      __syncthreads();
      QSTAMP(5);  // Y: barrier
      if (more) {
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = sh[par][11 + a][lane];
      }
      QKEEP(v[5]);
      QSTAMP(6);  // Y: LDS read of X's results
    }
  }
#ifdef QILQR_STAMPS
  stamp_flush();
#endif
}

// ---------------------------------------------------------------------------------------------
// k_rollout16: SIXTEEN LANES PER TRAJECTORY (rollout16.h): block = 192 = control wavefront A + pose wavefront B +
// operand wavefront P for four trajectories (b0 = 4 blockIdx.x; row r of 16 lanes <-> trajectory b0 + r).  At B = 1024
// that is 256 blocks -- one per CU -- instead of the 16 blocks of k_rollout3, and about a third of its instructions
// per knot.  The recurrence spans two knots (Log_i -> u_i -> v_{i+1} -> E_{i+1} -> T_{i+2} -> Log_{i+2}), so two knots can be
// in flight, and they are given to two IDENTICAL wavefronts one knot apart rather than to roles:
//   X_p (r16_wave_X; p = 0, 1): the whole step of the knots of parity p -- tau_i = Log(T_nom^-1 T_i),
//      u_i = u_nom + alpha k + K [tau_i ; v_i - v_nom], v_{i+1} = v_i + dt a(q_i, v_i, u_i), E_{i+1} = Exp(dt v_{i+1}),
//      T_{i+2} = T_{i+1} E_{i+1} -- from its own T_i and the other wave's v_i and T_{i+1}; stores knot i
//   P (r16::p_load, p_compute): for knot k (running up to R16_RING - 1 knots ahead): loads the nominal knot and the
//      gains (tiled global layout, per-lane element offsets), forms the 23 operand registers and writes them to ring slot
//      k % R16_RING as [register][lane]; copies the time column to the output trajectory.
// The waves never meet at a barrier inside the loop.  LDS words carry progress: knots produced by P, "v_k ready" (which
// also frees knot k - 1's operand slot), "T_k ready"; the values themselves go through four-deep LDS slots (k & 3).
// LDS operations of a wavefront execute in order, so a flag written after the data is seen after the data; every spin
// is bounded, so a lost flag ends the kernel instead of hanging it.
// Trajectories of the block that are not being rolled out this round alias the block's first live trajectory (their
// rows compute a duplicate that nobody stores): no row wanders onto a slow branch, no extra memory traffic.
// S = storage precision of trajectories and gains; the arithmetic is fp64 in either mode.
// ---------------------------------------------------------------------------------------------
struct DevWave {
  typedef double V;
  typedef bool M;
  typedef int I;
  template <class F> static __device__ __forceinline__ V vconst(F f) { return f((int)(threadIdx.x & 63)); }
  template <class F> static __device__ __forceinline__ M mconst(F f) { return f((int)(threadIdx.x & 63)); }
  // element indices of a knot become offsets into the tiled layout, (e / 2) * TILE2 + e % 2 (se3_math.h, knot_elem): the
  // loads of wavefront P are then a wave-uniform knot pointer plus a 32-bit lane offset, no address arithmetic per knot
  static __device__ __forceinline__ I iuni(int e) { return (e >> 1) * TILE2 + (e & 1); }
  template <class F> static __device__ __forceinline__ I iconst(F f) { return iuni(f((int)(threadIdx.x & 63))); }
  // value of lane L of the caller's row of 16 (v_mov_b64_dpp row_newbcast)
  template <int L> static __device__ __forceinline__ V bc(V x) { return __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, false); }
  // acc + x[lane L of the row] * m
  template <int L> static __device__ __forceinline__ V fm(V acc, V src, V m) {
    return __builtin_fma(bc<L>(src), m, acc);
  }
  // acc + sum_c src[lane L0 + c of the row] * m_c as a chain of v_fmac_f64_dpp (one instruction per term; the compiler
  // itself emits v_mov_b64_dpp + v_fma_f64, two).  The compiler's hazard recogniser does not look inside the asm, so the block
  // carries its own wait states on both sides: a DPP read needs two behind the VALU write of its source -- `src` may have
  // just been written, and `acc` may be the source of a DPP read right after.
#define QILQR_FMAC_DPP(m, l) "v_fmac_f64_dpp %0, %1, " m " row_newbcast:" l " row_mask:0xf bank_mask:0xf\n\t"
  template <int L0> static __device__ __forceinline__ V dot2(V acc, V src, V m0, V m1) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%4") QILQR_FMAC_DPP("%3", "%5") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "n"(L0), "n"(L0 + 1));
    return acc;
  }
  template <int L0> static __device__ __forceinline__ V dot3(V acc, V src, V m0, V m1, V m2) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%5") QILQR_FMAC_DPP("%3", "%6") QILQR_FMAC_DPP("%4", "%7") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "v"(m2), "n"(L0), "n"(L0 + 1), "n"(L0 + 2));
    return acc;
  }
  template <int L0> static __device__ __forceinline__ V dot4(V acc, V src, V m0, V m1, V m2, V m3) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%6") QILQR_FMAC_DPP("%3", "%7") QILQR_FMAC_DPP("%4", "%8") QILQR_FMAC_DPP("%5", "%9") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "n"(L0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
    return acc;
  }
#undef QILQR_FMAC_DPP
  // permutation inside every quad of four lanes (two v_mov_b32_dpp quad_perm: fp64 DPP has row_newbcast only)
  template <int CTRL> static __device__ __forceinline__ V qperm(V x) {
    const long long v = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  static __device__ __forceinline__ V rot1(V x) { return qperm<0xC9>(x); }  // lane j <- j + 1 (mod 3), j = 3 stays
  static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ bool any(M m) { return __ballot(m) != 0ull; }
  static __device__ __forceinline__ V sel(M m, V a, V b) { return m ? a : b; }
  static __device__ __forceinline__ M gt(V a, V b) { return a > b; }
  static __device__ __forceinline__ M lt(V a, V b) { return a < b; }
  static __device__ __forceinline__ M land(M a, M b) { return a && b; }
  static __device__ __forceinline__ M lor(M a, M b) { return a || b; }
  static __device__ __forceinline__ M lnot(M a) { return !a; }
  static __device__ __forceinline__ V abs_(V a) { return fabs(a); }
  static __device__ __forceinline__ V sqrt_(V a) { return sqrt(a); }
  static __device__ __forceinline__ V sin_(V a) { return sin(a); }
  static __device__ __forceinline__ V cos_(V a) { return cos(a); }
  static __device__ __forceinline__ V atan2_(V a, V b) { return atan2(a, b); }
  // the closed forms beyond the series' ranges: rarely taken and large (sin, cos, atan2 in fp64), so out of line -- the knot
  // loops stay short and the closed forms' registers are not the loops'
  static __device__ __attribute__((noinline)) V exp_closed(M c, V x, V p, M l0, M l1, M l2, M l3) {
    return r16::exp_closed_forms<DevWave>(c, x, p, l0, l1, l2, l3);
  }
  static __device__ __attribute__((noinline)) V log_closed(M c, V s2, V wq, V coeff) { return r16::log_closed_forms<DevWave>(c, s2, wq, coeff); }
  static __device__ __attribute__((noinline)) V jinv_closed(M c, V th2, V cJ) { return r16::jinv_closed_forms<DevWave>(c, th2, cJ); }
};

// s_waitcnt vmcnt(0) as an instruction the compiler's wait-count pass sees (gfx9 encoding: vmcnt in bits 3:0 and 15:14,
// expcnt 6:4 = 7 and lgkmcnt 11:8 = 15 left open).  Placed after the loads of a role's initial state: otherwise the pass may
// keep "a load is outstanding" alive around the knot loop and wait for vmcnt(0) INSIDE it -- which, the counter being shared,
// also waits for the knot's own stores, every knot.
#define R16_LOADS_DONE() __builtin_amdgcn_s_waitcnt(0x0F70)
constexpr int R16_RING = 4;
constexpr int R16_SPIN_MAX = 1 << 22;
#ifdef QILQR_DIAG
// diagnostics build: the knot whose velocity hand-off a step wavefront withholds (-1: none), so that the other wavefront's
// bounded spin runs out and the block's abort path is taken (tests/test_gpu_robustness.py)
__device__ int g_r16_stall_knot = -1;
#endif
constexpr int R16_CHUNK = 16;  // knots per "stored and visible" announcement of wave A (k_solve4's linearisation follows it)
enum { R16_F_PROD = 0, R16_F_V, R16_F_T, R16_F_K0, R16_F_K1, R16_F_ABORT, R16_NFLAGS };
enum { X_V = 0, X_T = 1 };
// LDS of the three rollout roles
struct R16Lds {
  double ops[R16_RING][r16::NOPS][64];  // operand registers of R16_RING knots, [register][lane]
  double xch[2][4][2][64];             // hand-off slots [X_V (v_lin, omega) of knot k | X_T (t, q) of knot k][k & 3][register][lane]
  int flags[R16_NFLAGS];               // knots produced by P; v_k ready (and knot k - 1's operands used); T_k ready; even, odd knots stored; abort
};
// The LDS executes the operations of one wavefront in the order they were issued, so a flag written after the data (or
// after the reads of a slot) is seen after them: no s_waitcnt, and no workgroup fence -- a release fence would wait for the
// wavefront's outstanding GLOBAL loads and stores too (vmcnt(0)), i.e. for P's prefetch and the knot stores, on every
// knot.  The asm statements only keep the compiler from moving LDS accesses across the flag.
__device__ __forceinline__ int r16_flag_read(R16Lds &sh, int which) {
  return __hip_atomic_load(&sh.flags[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void r16_flag_post(R16Lds &sh, int which, int value, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(&sh.flags[which], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// false: the flag never came (a bounded spin: the kernel ends instead of hanging); the abort word tells the other roles
__device__ __forceinline__ bool r16_flag_wait(R16Lds &sh, int which, int target, int seen, int lane) {
  if (__builtin_expect(seen >= target, 1)) return true;  // already observed (read ahead, one knot ago)
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    if (r16_flag_read(sh, which) >= target) {
      asm volatile("" ::: "memory");
      return true;
    }
    if ((spins & 255) == 255 && r16_flag_read(sh, R16_F_ABORT)) break;
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// the same for a wavefront that is in no hurry (the linearisation waits a chunk of knots at a time): sleeps between polls
__device__ __forceinline__ bool r16_flag_wait_relaxed(R16Lds &sh, int which, int target, int lane) {
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    if (r16_flag_read(sh, which) >= target) {
      asm volatile("" ::: "memory");
      return true;
    }
    if (r16_flag_read(sh, R16_F_ABORT)) return false;
    __builtin_amdgcn_s_sleep(16);
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// A hand-off read in ONE LDS round trip: the flag and the NV values are requested back to back (the LDS serves a
// wavefront's requests in order, so values read after a flag that shows `target` are the published ones); if the flag is not
// there yet, poll and read again.
template <int NV>
__device__ __forceinline__ bool r16_read_handoff(R16Lds &sh, int which, int target, int kind, int par, double (&d)[NV], int lane) {
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    asm volatile("" ::: "memory");  // read again, every time round
    const int f = r16_flag_read(sh, which);
    double a[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) a[r] = sh.xch[kind][par][r][lane];
    if (__builtin_expect(f >= target, 1)) {
#pragma unroll
      for (int r = 0; r < NV; ++r) d[r] = a[r];
      asm volatile("" ::: "memory");
      return true;
    }
    if ((spins & 255) == 255 && r16_flag_read(sh, R16_F_ABORT)) break;
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// the first attempt of r16_read_handoff split off, so that its LDS latency can be covered by other work: request the flag
// and the values here, do the other work, then r16_handoff_finish (which polls only if the first attempt came too early)
template <int NV>
__device__ __forceinline__ void r16_handoff_request(R16Lds &sh, int which, int kind, int par, int &f, double (&a)[NV], int lane) {
  asm volatile("" ::: "memory");
  f = r16_flag_read(sh, which);
#pragma unroll
  for (int r = 0; r < NV; ++r) a[r] = sh.xch[kind][par][r][lane];
  asm volatile("" ::: "memory");
}
template <int NV>
__device__ __forceinline__ bool r16_handoff_finish(R16Lds &sh, int which, int target, int kind, int par, int f, double (&a)[NV], int lane) {
  if (__builtin_expect(f >= target, 1)) return true;
  return r16_read_handoff<NV>(sh, which, target, kind, par, a, lane);
}
// "my knots up to i are stored and visible to the block": every fourth knot of its own a step wave waits for ALL its
// outstanding vector-memory operations (s_waitcnt vmcnt(0)) and announces the knots it has stored so far.  (Round 2 waited
// for all but the eight youngest operations -- vmcnt(8), "the stores of knots i .. i - 6 may be in flight" -- which is only
// right while a step issues exactly two stores and nothing else that counts: a spill inside the loop would have made the
// announcement early and the followers read knots not yet written, silently.  The full wait costs one store latency per
// eight knots on a path that only k_solve4 takes.)
__device__ __forceinline__ void r16_publish_stores(R16Lds &sh, int which, int i, int last, int lane) {
  if (i == last) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    r16_flag_post(sh, which, i + 1, lane);
  } else if (((i >> 1) & 3) == 3 && i >= 16) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    r16_flag_post(sh, which, i + 1, lane);
  }
}

// P: operand registers.  traj / gains: the nominal trajectory and the gains of the lane's trajectory (tiled); out: its
// candidate trajectory (the time column is copied there when `live`).
template <typename S>
__device__ __forceinline__ void r16_wave_P(R16Lds &sh, const S *traj, const S *gains, S *out, double alpha, bool live, int n, int lane,
                                           unsigned long long *stamps_out) {
  using namespace r16;
  PConsts<DevWave> pc;
  make_pconsts(pc);
  // The loads of knot k + 1 are requested before knot k is converted and written (two register sets, loop unrolled
  // by two: no copies).  A knot is 30 loads per lane: a wave-uniform knot pointer plus the lane's 32-bit offset.
  // The requests are unconditional (past the end the last knot is requested again): a branch around them makes the
  // compiler wait for vmcnt(0) at every use, i.e. for the requests it has just issued.
  S rawA[NRAW], rawB[NRAW], tmA, tmB;
  auto request = [&](int k, S (&raw)[NRAW], S &tm) {
    const int kk = k < n ? k : n - 1;
    const S *tk = traj + (long)kk * (9 * TILE2), *gk = gains + (long)kk * (26 * TILE2);
    tm = tk[0];  // time_s: the oldest request of the knot
    auto ld = [&](int off) -> S { return tk[off]; };
    auto lg = [&](int off) -> S { return gk[off]; };
    p_load<DevWave>(pc, ld, lg, raw);
  };
  bool ok = true;
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  auto knot = [&](int k, S (&rc)[NRAW], S &tmc, S (&rn)[NRAW], S &tmn) {
    request(k + 1, rn, tmn);
    QSTAMP(0);  // P: requests
    double op[NOPS];
    p_compute<DevWave>(pc, rc, alpha, op);
    QKEEP(op[0]); QKEEP(op[r16::NOPS - 1]); QKEEP(op[10]);
    QSTAMP(1);  // P: wait for the loads, operand registers
    // slot k % R16_RING is free once the step of knot k - R16_RING has used its operands: that step posts v_{k - R16_RING + 1}
    // behind their last use, and the "v ready" word only grows (a step posts after it has taken the previous step's v)
    if (k >= R16_RING && !r16_flag_wait(sh, R16_F_V, k - R16_RING + 1, -1, lane)) ok = false;
    QSTAMP(2);  // P: wait for a free slot
#pragma unroll
    for (int r = 0; r < NOPS; ++r) sh.ops[k % R16_RING][r][lane] = op[r];
    r16_flag_post(sh, R16_F_PROD, k + 1, lane);
    if (live && (lane & 15) == 0) out[knot_elem<true>(k, 0, 18)] = tmc;  // time_s passes through (ilqr.hh:164)
    QSTAMP(3);  // P: LDS writes, flag, time store
  };
  request(0, rawA, tmA);
  for (int k = 0; k < n && ok; k += 2) {
    knot(k, rawA, tmA, rawB, tmB);
    if (k + 1 < n && ok) knot(k + 1, rawB, tmB, rawA, tmA);
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
}

// X_p: the steps of the knots of parity p.  Step i, from T_i (this wave's own, out of its step i - 2):
//     tau_i = Log(T_nom^-1 T_i);  u_i;  v_{i+1} = v_i + dt a(q_i, v_i, u_i)  -> handed to the other wave;
//     E_{i+1} = Exp(dt v_{i+1});  T_{i+2} = T_{i+1} E_{i+1}                   -> handed to the other wave, and kept
// with v_i and T_{i+1} from the other wave's step i - 1.  The recurrence spans two knots, so two such waves, one knot apart,
// never wait for each other in the steady state: v_i is posted about half a step before step i needs it, T_{i+1} likewise --
// the hand-offs' LDS latency (~450 cycles from post to use through a progress word, which bounded every partition of a
// knot into roles: 74 us) is off the chain.  (TT, QQ, VL, VW): the state of knot 0.  PUBLISH: announce the stored knots.
template <typename S, bool PUBLISH>
__device__ __forceinline__ void r16_wave_X(R16Lds &sh, const ModelConsts<double> &c, int p, double TT, double QQ, double VL, double VW, S *out,
                                           bool live, int n, int lane, unsigned long long *stamps_out) {
  using namespace r16;
  RConsts<DevWave> kc;
  make_rconsts(c, kc);
  const int ea = sta_elem(lane), ep = stp_elem(lane);
  const bool wa = live && ea >= 0, wp = live && ep >= 0;
  const int oa = DevWave::iuni(ea >= 0 ? ea : 0), opz = DevWave::iuni(ep >= 0 ? ep : 0);
  const int last = (n - 1) - (((n - 1) & 1) ^ p);  // this wave's last knot (< 0: none)
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  if (p == 1 && n > 1) {  // "step -1": T_1 = T_0 Exp(dt v_0), this wave's first pose and the even wave's first compose
    double DQ, PP, TTn, QQn;
    a_exp<DevWave>(kc, VL, VW, DQ, PP);
    b_compose<DevWave>(kc, TT, QQ, DQ, PP, TTn, QQn);
    TT = TTn;
    QQ = QQn;
    sh.xch[X_T][1][0][lane] = TT;
    sh.xch[X_T][1][1][lane] = QQ;
    r16_flag_post(sh, R16_F_T, 1, lane);
  }
  int seen = -1;  // P's progress as last read
  // One step.  `op`: the knot's operand registers, already requested by the previous step of this wave if `have` (their LDS
  // round trip then lies behind that step's compose); `opn` / `have_n`: the same for this wave's next knot, requested here.
  // The loop calls it with the two register sets alternating (no copies).
  auto step = [&](int i, double (&op)[NOPS], bool have, double (&opn)[NOPS], bool &have_n) -> bool {
    if (!have) {
      if (__builtin_expect(!r16_flag_wait(sh, R16_F_PROD, i + 1, seen, lane), 0)) return false;
#pragma unroll
      for (int r = 0; r < NOPS; ++r) op[r] = sh.ops[i % R16_RING][r][lane];
      seen = r16_flag_read(sh, R16_F_PROD);
    }
    have_n = false;
    S *ok_ = out + (long)i * (9 * TILE2);
    if (wp) ok_[opz] = (S)__builtin_fma(kc.MQ0, QQ, TT);  // [q | t] in one store
    QSTAMP(0);  // X: operand reads (if not requested ahead), pose store
    double TH4, TD;
    b_log<DevWave>(kc, TT, QQ, op, TH4, TD);
    // v_i: the other wave posts it about now (after its control section, half a step ahead of this one's): requested as late as
    // the LDS round trip allows -- a request that comes before the post has to be repeated by polling
    int fv = 0;
    double v[2] = {VL, VW};
    if (i > 0) r16_handoff_request<2>(sh, R16_F_V, X_V, i & 3, fv, v, lane);
    const double RH = a_rho<DevWave>(TH4, TD);
    QSTAMP(1);  // X: Log
    if (i > 0) {
      if (__builtin_expect(!r16_handoff_finish<2>(sh, R16_F_V, i, X_V, i & 3, fv, v, lane), 0)) return false;
    }
    QSTAMP(2);  // X: wait for v_i
    const bool advance = i + 1 < n;  // the reference's step after the last knot is computed and discarded (ilqr.hh:168)
    APre<DevWave> pre;
    a_pre<DevWave>(kc, v[0], v[1], op, pre);
    double VLn = 0.0, VWn = 0.0;
    const double st = a_post<DevWave, true>(kc, pre, TH4, RH, QQ, v[0], op, advance, VLn, VWn);
    if (__builtin_expect(advance, 1)) {  // (the post also tells P that knot i's operand slot is free)
      sh.xch[X_V][(i + 1) & 3][0][lane] = VLn;
      sh.xch[X_V][(i + 1) & 3][1][lane] = VWn;
#ifdef QILQR_DIAG
      if (i != g_r16_stall_knot)  // fault injection (qilqr_debug_set_rollout_stall): this hand-off is never announced
#endif
      r16_flag_post(sh, R16_F_V, i + 1, lane);
    }
    if (wa) ok_[oa] = (S)st;
    if (PUBLISH) r16_publish_stores(sh, R16_F_K0 + p, i, last, lane);
    QSTAMP(3);  // X: control, velocity, hand-off, store
    if (i + 2 < n) {
      int ft;
      double t[2];
      r16_handoff_request<2>(sh, R16_F_T, X_T, (i + 1) & 3, ft, t, lane);  // T_{i+1}: posted at the end of the other wave's step i - 1
      double DQ, PP;
      a_exp<DevWave>(kc, VLn, VWn, DQ, PP);
      QSTAMP(4);  // X: Exp
      if (__builtin_expect(!r16_handoff_finish<2>(sh, R16_F_T, i + 1, X_T, (i + 1) & 3, ft, t, lane), 0)) return false;
      QSTAMP(5);  // X: wait for T_{i+1}
      // the operands of this wave's next knot, if P has them (it is normally three knots ahead): requested here, used after
      // the compose
      if (seen >= i + 3) {
#pragma unroll
        for (int r = 0; r < NOPS; ++r) opn[r] = sh.ops[(i + 2) % R16_RING][r][lane];
        seen = r16_flag_read(sh, R16_F_PROD);
        have_n = true;
      }
      double TTn, QQn;
      b_compose<DevWave>(kc, t[0], t[1], DQ, PP, TTn, QQn);
      TT = TTn;
      QQ = QQn;
      sh.xch[X_T][(i + 2) & 3][0][lane] = TT;
      sh.xch[X_T][(i + 2) & 3][1][lane] = QQ;
      r16_flag_post(sh, R16_F_T, i + 2, lane);
      QSTAMP(6);  // X: compose, hand-off
    }
    return true;
  };
  double opA[NOPS], opB[NOPS];
  bool haveA = false, haveB = false;
  for (int i = p; i < n; i += 4) {
    if (!step(i, opA, haveA, opB, haveB)) return;
    if (i + 2 >= n) break;
    if (!step(i + 2, opB, haveB, opA, haveA)) return;
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
}

template <typename S>
__global__ __launch_bounds__(192) void k_rollout16(ModelConsts<double> c, BatchState st, int B, int n, int need_flag) {
#define R16_RETURN return
#include "rollout16_body.inc"
#undef R16_RETURN
}
// k_backward_rollout: the two in ONE launch for batches whose blocks of four trajectories all fit the chip at once (one block
// per CU: the rollout's 214 registers): the block's backward pass (fused, barrier-free form), a block barrier, then the
// rollout of its own four trajectories by wavefronts 0..2 -- gains, flags, step sizes written and read by the same CU.  One
// launch boundary and one kernel start fewer per round.
template <typename S>
__global__ __launch_bounds__(320) void k_backward_rollout(ModelConsts<double> c, SolveParams p, BatchState st, int B, int n) {
  {
    constexpr int WAVES = 5;
    constexpr bool FUSED = true, FREE = true;
    const int force = 0;
    (void)WAVES;
#define BW4_RETURN goto backward_done
#include "backward4_body.inc"
#undef BW4_RETURN
  }
backward_done:
  __syncthreads();  // (every wavefront comes out of the backward pass; its stores are visible to the block)
  if (threadIdx.x >= 192) return;
  {
    const int need_flag = F_SEARCH;
#define R16_RETURN return
#include "rollout16_body.inc"
#undef R16_RETURN
  }
}

// ---------------------------------------------------------------------------------------------
// k_round: a whole round in ONE launch, where k_backward_rollout is allowed (a block per CU) -- its backward pass and rollout, then the
// linearisation of the block's own candidates by all five wavefronts (k_linearize's arithmetic: se3_math.h forms its fused
// multiply-adds from the source alone, so the records are the same bits whichever kernel writes them).  One launch boundary and one
// kernel start and end fewer per round than k_backward_rollout + k_linearize.  The count of running trajectories is complete only
// when every block has settled, so a launch hands the host the count of the round BEFORE it (prev_counters, prev_round): the host
// alternates between two sets of counters, and an idle wavefront of block 0 publishes while the others roll out.
// ---------------------------------------------------------------------------------------------
// k_linearize's work for the trajectories of one block: lane-tasks (knot, live trajectory) of the cost half first -- the longer
// chain -- then of the dynamics half, sixty-four to a wavefront, wavefront w of the block taking tasks w, w + nwaves, ...
template <typename S, int LK>
__device__ __forceinline__ void linearize_block(const ModelConsts<S> &c, const S *qr, const BatchState &st, int b0, int B, int n, int which,
                                                int need_flag) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int)(blockDim.x >> 6);
  int l0 = 0, l1 = 0, l2 = 0, l3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0, nl = 0;  // slots and buffers of the trajectories that take part
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int b = b0 + g;
    if (b >= B) continue;
    const int fl = st.flags[b], buf = st.cur[b] ^ which;
    if (need_flag && !(fl & need_flag)) continue;
    if (nl == 0) { l0 = b; u0 = buf; } else if (nl == 1) { l1 = b; u1 = buf; } else if (nl == 2) { l2 = b; u2 = buf; } else { l3 = b; u3 = buf; }
    ++nl;
  }
  if (nl == 0) return;
  const int per = nl * n, wt = (per + 63) >> 6;
  for (int t = wave; t < 2 * wt; t += nwaves) {
    const bool cost_half = t < wt;  // wave-uniform
    const int r = (cost_half ? t : t - wt) * 64 + lane;
    if (r >= per) continue;
    const int i = (nl == 1) ? r : (nl == 2) ? (r >> 1) : (nl == 3) ? r / 3 : (r >> 2);
    const int g = r - i * nl;
    const int b = (g == 0) ? l0 : (g == 1) ? l1 : (g == 2) ? l2 : l3;
    const int buf = (g == 0) ? u0 : (g == 1) ? u1 : (g == 2) ? u2 : u3;
    S pt[18];
    load_knot<true>((const S *)st.traj[buf] + knot_base<true>(b, n, 18), i, 18, pt);
    S *rec = (S *)st.lin[buf] + rec_base(st.layout, b, n) + rec_elem(st.layout, i, 0);
    if (!cost_half) {
      const TiledRecWriter<S> wd{rec};
      linearize_dynamics(c, pt, wd);
      wd.flush();
    } else {
      const TiledRecWriter<S> w{rec};
      S pd[18];
      if (st.desired_tiled) load_knot<true>((const S *)st.desired + knot_base<true>(b, n, 18), i, 18, pd);
      else load_knot<false>((const S *)st.desired, i, 18, pd);
      const S cost = linearize_cost<LK>(qr, qr + 144, pt, pd, w);
      w.flush();
      st.knot_cost[buf][cost_index(b, i, n)] = (double)cost;
    }
  }
}
// the count of running trajectories of a finished round to the host (k_linearize's first wavefront does the same for its own round)
__device__ __forceinline__ void publish_active(int *counters, unsigned long long *host_active, int round, int lane) {
  int act = counters[COUNT_BASE + lane];
  counters[COUNT_BASE + lane] = 0;  // (the round after next counts into these words again)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) act += __shfl_xor(act, off);
  if (lane == 0)
    __hip_atomic_store(&host_active[round & 7], ((unsigned long long)(unsigned)(round + 1) << 32) | (unsigned)act, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_publish_active(int *counters, unsigned long long *host_active, int round) {
  publish_active(counters, host_active, round, threadIdx.x & 63);
}
// ROUNDS > 1: several rounds per launch (a round's settle step finds the knot costs the block has just written): the launch
// boundaries between them are gone too.  All of them count into the launch's counter set: the host reads the SUM of their counts of
// running trajectories, an upper bound of the last one's and zero exactly when the first one's is.  The rounds share the block's LDS.
template <int LK, int ROUNDS>
__global__ __launch_bounds__(320) void k_round(ModelConsts<double> c, const ModelConsts<double> *__restrict__ cp, SolveParams p, BatchState st, int B,
                                                int n, int *prev_counters, int prev_round) {
  typedef double S;
  __shared__ double qr_w[160];  // the weights of the cost half (k_linearize keeps a copy per wavefront: here the block's)
  BW4_DECLARE_LDS
  __shared__ R16Lds sh;
#define BW4_LDS_DECLARED
#define R16_LDS_DECLARED
#define BW4_CTAB_FILLED
  for (int k = threadIdx.x; k < 160; k += blockDim.x) qr_w[k] = (k < 144) ? cp->Q[k] : cp->R[k - 144];
  // the constant operand table behind the ring slots, once for all the rounds of the launch (a round whose block has nothing to run
  // leaves before it would fill it, and a later round of the same launch may have something: so here, unconditionally; the records and
  // the settle step's scratch use the slots' other words)
  bw4_fill_ctab<S>(ring, st.ctab, 320);
  // (an idle wavefront of block 0 hands the host the count of the launch before this one while the others roll out)
#define ROUND_BEHIND_BACKWARD \
  if (blockIdx.x == 0 && (threadIdx.x >> 6) == 4 && prev_round >= 0) publish_active(prev_counters, st.host_active, prev_round, threadIdx.x & 63);
#define ROUND_ID 0
#include "round_body.inc"
#undef ROUND_ID
#undef ROUND_BEHIND_BACKWARD
#define ROUND_BEHIND_BACKWARD
  if constexpr (ROUNDS > 1) {
#define ROUND_ID 1
#include "round_body.inc"
#undef ROUND_ID
  }
  if constexpr (ROUNDS > 2) {
#define ROUND_ID 2
#include "round_body.inc"
#undef ROUND_ID
#define ROUND_ID 3
#include "round_body.inc"
#undef ROUND_ID
  }
#undef ROUND_BEHIND_BACKWARD
#undef BW4_LDS_DECLARED
#undef R16_LDS_DECLARED
#undef BW4_CTAB_FILLED
}

// ---------------------------------------------------------------------------------------------
// k_accept: thread b.  Cost of the candidate, acceptance, convergence (ilqr.hh:70-84, 174-194)
// ---------------------------------------------------------------------------------------------
__global__ void k_accept(SolveParams p, BatchState st, int B, int n, int ls_only) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int fl = st.flags[b];
  if (fl & F_SEARCH) {
    const int cur = st.cur[b];
    const double *kc = st.knot_cost[cur ^ 1];
    double new_cost = 0.0;
    for (int i = 0; i < n; ++i) new_cost += kc[cost_index(b, i, n)];
    st.n_fwd[b] += 1;
    const int it = st.iters[b];
    const double cost = st.prev_cost[b];
    const double alpha = st.alpha[b];
    bool accept;
    if (it == 0) {
      accept = true;  // ilqr.hh:71-73: the first rollout is taken unconditionally
    } else {
      const double desired = p.reduction_frac * cost_reduction(st.terms[2 * b], st.terms[2 * b + 1], alpha);
      accept = (new_cost - cost < desired);  // ilqr.hh:186
    }
    if (accept && ls_only) {
      // stand-alone ILQR::line_search: report the accepted candidate, no outer-loop bookkeeping
      st.cur[b] = cur ^ 1;
      st.cost[b] = new_cost;
      st.status[b] = 0;
      fl = 0;
    } else {
      // (values first, then the stores of store_settled: no store in one branch that a store of the other complements)
      const int trial0 = st.trial[b];
      int status = -1;
      if (accept) {
        st.mu[b] = lm_relax(p, st.mu[b]);
        fl = F_ACTIVE;
        if (it > 0 && is_converged(p, cost, new_cost)) {
          status = 1;  // ilqr.hh:82-84
          fl = 0;
        } else if (!((double)(it + 1) < p.max_iters)) {
          status = 2;  // ilqr.hh:86
          fl = 0;
        }
      } else if (trial0 + 1 >= p.ls_max_iters) {
        double mu = (p.mu_init > 0.0) ? st.mu[b] : 0.0;
        if (!ls_only && lm_restart(p, mu)) {
          st.mu[b] = mu;
          fl = F_ACTIVE;  // the next backward pass runs on the same iterate with the larger mu
        } else {
          status = 3;  // ilqr.hh:191-193
          fl = 0;
        }
      }
      store_settled(st, b, accept, accept ? cur ^ 1 : cur, new_cost, it, trial0, alpha, p.step_update, status, fl);
    }
    st.flags[b] = fl;
  }
  if (fl & F_ACTIVE) atomicAdd(active_counter(st), 1);
}

// ---------------------------------------------------------------------------------------------
// k_gather: results into caller buffers in the plain [B][n][18] layout (any may be null).
// Block (b, chunk): trajectory b = blockIdx.x, thread = one 16-byte entry pair of it (no 64-bit division per element:
// with one thread per element and three of them this kernel and k_retile took 13 and 16 us for 14.7 MB each)
// ---------------------------------------------------------------------------------------------
// mask (optional): only the trajectories with mask[b] == want take part.  row_of (optional): trajectory b goes to row row_of[b]
// of the output arrays instead of row b (the compact copy of the trajectories that finished late, k_late_slots).
template <typename S>
__global__ void k_gather(BatchState st, int B, int n, double *out_traj, double *out_cost, int *out_status,
                         int *out_iters, int *out_n_bwd, int *out_n_fwd, const int *mask, int want, const int *row_of) {
  // thread = one 16-byte piece of tile blockIdx.x in the order the tile is stored: (knot, pair) q >> TILE_LOG of trajectory
  // slot q & (TILE - 1) -- the tiled side is one contiguous run per wavefront, the plain side TILE runs
  const int q = blockIdx.y * blockDim.x + threadIdx.x;
  const int b = blockIdx.x * TILE + (q & (TILE - 1)), kp = q >> TILE_LOG;
  if (kp >= n * 9 || b >= B) return;
  if (mask && mask[b] != want) return;
  const long row = row_of ? row_of[b] : b;
  if (row < 0) return;  // (compaction: the slot's trajectory has moved to another slot, or left through k_compact_move)
  if (out_traj) {
    typedef typename GA<S>::v2 sv2;
    const int i = kp / 9, pr = kp - 9 * i;
    const sv2 v = *reinterpret_cast<const sv2 *>((const S *)st.traj[st.cur[b]] + knot_base<true>(b, n, 18) + knot_elem<true>(i, 2 * pr, 18));
    double *o = out_traj + (row * n * 9 + kp) * 2;
    o[0] = (double)v.x;
    o[1] = (double)v.y;
  }
  if (kp == 0) {
    if (out_cost) out_cost[row] = st.cost[b];
    if (out_status) out_status[row] = st.status[b];
    if (out_iters) out_iters[row] = st.iters[b];
    if (out_n_bwd) out_n_bwd[row] = st.n_bwd[b];
    if (out_n_fwd) out_n_fwd[row] = st.n_fwd[b];
  }
}
// ---------------------------------------------------------------------------------------------
// Compaction of the live trajectories (round 4; large batches only -- the host decides, ilqr_capi.hip compaction_on).
// A batch takes as many rounds as its slowest problem (configs[3]: 45 for a mean of 12.4 iterations), and the kernels give
// out work in groups of slots -- k_backward4 four to a block, k_linearize and k_rollout3 sixty-four to a wavefront -- that cost
// the same with one live trajectory as with all: by round 15 a quarter of the trajectories are live and they still occupy
// 71 % of the blocks of four and every group of 64.  Between a round's backward pass (whose settle step is where a
// trajectory gets its exit status) and its rollout, the live trajectories are therefore moved into a dense prefix of the
// slots: k_compact_plan (one block) pairs the holes among the first L slots (L = the live count) with the live slots
// behind them; k_compact_move (one block per pair) first gathers the result of the hole's finished trajectory -- if it has
// one -- into the CALLER's arrays, then copies the live trajectory's state: its current trajectory, its gains, its scalars
// (and its knot records when Levenberg-Marquardt restarts are on: a restart runs the recursion on them again; otherwise the
// next records a live trajectory needs are the ones k_linearize is about to write).  Every kernel addresses a trajectory by
// its slot, none by its row: st.orig carries the row along and k_gather puts the results where they belong.  A trajectory
// moves at most once per round and only from behind the prefix into it, so over a solve at most B trajectories move.  The
// arithmetic of a trajectory does not depend on its slot: results are bit-identical with and without (GPU tests).
// ---------------------------------------------------------------------------------------------
constexpr int PLAN_HEAD = 16;
// inclusive scan over the 1024 threads of a block (sixteen wavefronts): shuffles inside a wavefront, the sixteen totals through LDS
template <typename T>
__device__ __forceinline__ T block_scan_1024(T v, T (&tot)[16], T *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T u = __shfl_up(v, off);
    if (lane >= off) v += u;
  }
  __syncthreads();  // (tot may still be read from the previous scan)
  if (lane == 63) tot[w] = v;
  __syncthreads();
  T base = 0, all = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const T x = tot[k];
    if (k < w) base += x;
    all += x;
  }
  *total = all;
  return v + base;
}
__global__ __launch_bounds__(1024) void k_compact_plan(BatchState st, int B) {
  __shared__ int s_tot[16];
  __shared__ unsigned long long s_tot2[16];
  const int t = threadIdx.x;
  // a thread's flags are consecutive words, a multiple of four of them, read sixteen bytes at a time (at most 64 words for 65536
  // slots: one word per load took this kernel 80 us there); they stay in the L2 between the passes
  const int per = (((B + 1023) / 1024) + 3) & ~3;
  const int b0 = t * per < B ? t * per : B, b1 = (b0 + per < B) ? b0 + per : B;
  auto alive4 = [&](int b, bool (&al)[4]) {  // slots b .. b + 3 (b a multiple of four; beyond B: not alive)
    if (b + 3 < B) {
      const int4 v = *reinterpret_cast<const int4 *>(st.flags + b);
      al[0] = (v.x & F_ACTIVE) != 0; al[1] = (v.y & F_ACTIVE) != 0; al[2] = (v.z & F_ACTIVE) != 0; al[3] = (v.w & F_ACTIVE) != 0;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) al[e] = (b + e < B) && (st.flags[b + e] & F_ACTIVE) != 0;
    }
  };
  int live = 0;
  for (int b = b0; b < b1; b += 4) {
    bool al[4];
    alive4(b, al);
    live += (int)al[0] + (int)al[1] + (int)al[2] + (int)al[3];
  }
  int L;
  (void)block_scan_1024(live, s_tot, &L);
  // holes among the first L slots (low word) and live slots behind them (high word), ranked in one scan
  unsigned long long hm = 0;
  for (int b = b0; b < b1; b += 4) {
    bool al[4];
    alive4(b, al);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (b + e >= B) continue;
      if (b + e < L) hm += al[e] ? 0ull : 1ull;
      else hm += al[e] ? (1ull << 32) : 0ull;
    }
  }
  unsigned long long all;
  const unsigned long long incl = block_scan_1024(hm, s_tot2, &all);
  int hk = (int)(unsigned)(incl - hm), mk = (int)((incl - hm) >> 32);  // exclusive ranks
  int *dst = st.plan + PLAN_HEAD, *src = st.plan + PLAN_HEAD + B;
  for (int b = b0; b < b1; b += 4) {
    bool al[4];
    alive4(b, al);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (b + e >= B) continue;
      if (b + e < L) { if (!al[e]) dst[hk++] = b + e; }
      else if (al[e]) src[mk++] = b + e;
    }
  }
  const int M = (int)(all >> 32);  // (as many holes in front of L as live slots behind it)
  if (t == 0) {
    st.plan[0] = M;
    st.plan[1] = L;
    st.plan[2] += M;
  }
  // what the parts of a pair must agree on, noted before any of them runs: the row of the hole's finished trajectory (-1: none),
  // its buffer selector, the live trajectory's selector
  __syncthreads();  // (dst and src of a pair were written by different threads; global writes of this block, visible behind the barrier)
  int *pair = st.plan + PLAN_HEAD + 2 * B;
  for (int k = t; k < M; k += 1024) {
    const int d = dst[k], sr = src[k];
    pair[4 * k] = st.orig[d];
    pair[4 * k + 1] = st.cur[d];
    pair[4 * k + 2] = st.cur[sr];
  }
}
struct CompactOut {  // the caller's result arrays (device pointers, any may be null): k_gather's
  double *traj, *cost;
  int *status, *iters, *n_bwd, *n_fwd;
};
// 16-byte pieces q0 <= q < q1 of a tiled (step = TILE2) or plain (step = 2) run, four loads in flight per thread
template <typename S>
__device__ __forceinline__ void copy_pieces(const S *a, S *b, long step, int q0, int q1) {
  typedef typename GA<S>::v2 sv2;
  const int t = threadIdx.x, nt = blockDim.x;
  int q = q0 + t;
  for (; q + 3 * nt < q1; q += 4 * nt) {
    sv2 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const sv2 *>(a + (long)(q + e * nt) * step);
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<sv2 *>(b + (long)(q + e * nt) * step) = v[e];
  }
  for (; q < q1; q += nt) *reinterpret_cast<sv2 *>(b + (long)q * step) = *reinterpret_cast<const sv2 *>(a + (long)q * step);
}
// Work item = (pair k, part c of COMPACT_SPLIT).  The units of a pair -- n * 9 pieces of trajectory (the hole's finished
// trajectory out to the caller's row, then the live one in: the same thread does both for a piece, in that order, because the
// hole's current trajectory may sit in the buffer the copy writes), n * 26 pieces of gains, and the records' -- are numbered
// through and cut into equal parts; part 0's first thread also moves the scalars.  The parts of a pair read the pair's slots and
// selectors as they were when the part started: k_compact_pairs (below) notes them in the plan before any part runs.
constexpr int COMPACT_SPLIT = 8;
template <typename S>
__global__ __launch_bounds__(256) void k_compact_move(BatchState st, int B, int n, CompactOut out, int with_records) {
  typedef typename GA<S>::v2 sv2;
  const int M = st.plan[0];
  const int t = threadIdx.x;
  const int *pair = st.plan + PLAN_HEAD + 2 * B;  // [k][4]: row of the hole's trajectory, its selector, the live one's selector
  const int nt = n * 9, ng = n * 26, nr = with_records ? n * (st.layout.stride / 2) : 0;
  const int units = nt + ng + nr;
  for (int item = blockIdx.x; item < M * COMPACT_SPLIT; item += gridDim.x) {
    const int k = item / COMPACT_SPLIT, c = item - k * COMPACT_SPLIT;
    const int dst = st.plan[PLAN_HEAD + k], src = st.plan[PLAN_HEAD + B + k];
    const long row = pair[4 * k];
    const int cd = pair[4 * k + 1], cs = pair[4 * k + 2];
    const int u0 = (int)((long)units * c / COMPACT_SPLIT), u1 = (int)((long)units * (c + 1) / COMPACT_SPLIT);
    // trajectory pieces [u0, u1) ∩ [0, nt)
    {
      const int q1 = u1 < nt ? u1 : nt;
      const S *dead = (const S *)st.traj[cd] + knot_base<true>(dst, n, 18);
      const S *a = (const S *)st.traj[cs] + knot_base<true>(src, n, 18);
      S *b = (S *)st.traj[cs] + knot_base<true>(dst, n, 18);
      double *o = (row >= 0 && out.traj) ? out.traj + row * n * 18 : nullptr;
      for (int q = u0 + t; q < q1; q += blockDim.x) {
        const sv2 live = *reinterpret_cast<const sv2 *>(a + (long)q * TILE2);
        if (o) {
          const sv2 v = *reinterpret_cast<const sv2 *>(dead + (long)q * TILE2);
          o[2 * q] = (double)v.x;
          o[2 * q + 1] = (double)v.y;
        }
        *reinterpret_cast<sv2 *>(b + (long)q * TILE2) = live;
      }
    }
    // gains pieces
    {
      const int g0 = (u0 > nt ? u0 : nt) - nt, g1 = (u1 < nt + ng ? u1 : nt + ng) - nt;
      if (g0 < g1) copy_pieces<S>((const S *)st.gains + knot_base<true>(src, n, 52), (S *)st.gains + knot_base<true>(dst, n, 52), TILE2, g0, g1);
    }
    if (nr) {
      const RecLayout &L = st.layout;
      const int r0 = (u0 > nt + ng ? u0 : nt + ng) - nt - ng, r1 = u1 - nt - ng;
      if (r0 < r1) copy_pieces<S>((const S *)st.lin[cs] + rec_base(L, src, n), (S *)st.lin[cs] + rec_base(L, dst, n), L.tiled ? TILE2 : 2, r0, r1);
    }
    if (c == 0 && t == 0) {
      if (row >= 0) {
        if (out.cost) out.cost[row] = st.cost[dst];
        if (out.status) out.status[row] = st.status[dst];
        if (out.iters) out.iters[row] = st.iters[dst];
        if (out.n_bwd) out.n_bwd[row] = st.n_bwd[dst];
        if (out.n_fwd) out.n_fwd[row] = st.n_fwd[dst];
      }
      st.cur[dst] = cs;
      st.cost[dst] = st.cost[src];
      st.prev_cost[dst] = st.prev_cost[src];
      st.terms[2 * dst] = st.terms[2 * src];
      st.terms[2 * dst + 1] = st.terms[2 * src + 1];
      st.alpha[dst] = st.alpha[src];
      st.mu[dst] = st.mu[src];
      st.trial[dst] = st.trial[src];
      st.status[dst] = st.status[src];
      st.iters[dst] = st.iters[src];
      st.n_bwd[dst] = st.n_bwd[src];
      st.n_fwd[dst] = st.n_fwd[src];
      st.orig[dst] = st.orig[src];
      st.flags[dst] = st.flags[src];
      st.flags[src] = 0;  // nothing runs in the slot it left, and k_gather passes it by
      st.orig[src] = -1;
    }
  }
}

// ILQRDebug on the device (ilqr.hh:78-80: one entry per completed forward pass, the accepted trajectory and its cost) for the
// single-problem solve: launched behind every round's backward pass (whose settle step is where an iteration completes), one
// block; when trajectory 0 has completed an iteration since the last look, its current trajectory -- in the buffer the next
// rollout does not write -- and cost go to row `seen` of the ring, plain [n][18] layout.  No host round trip: the rounds stay
// free-running and the ring is downloaded once, after the solve (round 4; round 3 synchronised every round and copied from
// the host).
template <typename S>
__global__ void k_debug_capture(BatchState st, int n, double *dbg_trajs, double *dbg_cost, int *dbg_seen, int cap) {
  __shared__ int s_seen;
  if (threadIdx.x == 0) s_seen = *dbg_seen;
  __syncthreads();
  const int seen = s_seen, it = st.iters[0];
  if (it <= seen) return;
  if (seen < cap) {
    typedef typename GA<S>::v2 sv2;
    const S *t = (const S *)st.traj[st.cur[0]] + knot_base<true>(0, n, 18);
    double *o = dbg_trajs ? dbg_trajs + (size_t)seen * n * 18 : nullptr;
    if (o)
      for (int kp = threadIdx.x; kp < n * 9; kp += blockDim.x) {
        const int i = kp / 9, pr = kp - 9 * i;
        const sv2 v = *reinterpret_cast<const sv2 *>(t + knot_elem<true>(i, 2 * pr, 18));
        o[2 * kp] = (double)v.x;
        o[2 * kp + 1] = (double)v.y;
      }
    if (threadIdx.x == 0 && dbg_cost) dbg_cost[seen] = st.cost[0];
  }
  if (threadIdx.x == 0) *dbg_seen = it;  // (every thread took `seen` from shared memory in front of the barrier's other side)
}
// The copy-back of a host-buffer batch solve in two parts (qilqr_solve_batch): k_mark_final, on the solver's stream between two
// rounds, notes which trajectories have reached their exit status (nothing of theirs changes any more); those are gathered
// and copied to the host on a second stream while the rounds of the others go on.  k_late_slots, after the last round, gives
// each of the others a row of a small compact buffer.
__global__ void k_mark_final(BatchState st, int B, int *early, int *late_count) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) *late_count = 0;
  if (b < B) early[b] = (st.flags[b] == 0) ? 1 : 0;
}
__global__ void k_late_slots(int B, const int *early, int *late_count, int *late_idx, int *late_slot, int cap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || early[b]) return;
  const int slot = atomicAdd(late_count, 1);
  late_slot[b] = slot < cap ? slot : cap - 1;  // (never more than cap: the count of running trajectories only falls)
  if (slot < cap) late_idx[slot] = b;
}

// plain [B][n][W] <-> tiled, W = 18 or 52 (even).  to_tiled = 1: plain -> tiled.  sel (optional): per-trajectory choice of
// tiled buffer t0 / t1 (the current-trajectory selector), xor'ed with flip.  Threads as in k_gather.
template <typename S>
__global__ void k_retile(const double *plain_in, double *plain_out, S *t0, S *t1, const int *sel,
                         int flip, int B, int n, int W, int to_tiled, int *zero_word) {
  const int q = blockIdx.y * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && q == 0 && zero_word) *zero_word = 0;  // the group queue of the k_solve4 launch that follows
  const int hw = W >> 1;
  const int b = blockIdx.x * TILE + (q & (TILE - 1)), kp = q >> TILE_LOG;
  if (kp >= n * hw || b >= B) return;
  typedef typename GA<S>::v2 sv2;
  const int i = kp / hw, pr = kp - hw * i;
  S *t = (sel && ((sel[b] ^ flip) & 1)) ? t1 : t0;
  sv2 *tp = reinterpret_cast<sv2 *>(t + (W == 18 ? knot_base<true>(b, n, 18) + knot_elem<true>(i, 2 * pr, 18)
                                                   : knot_base<true>(b, n, 52) + knot_elem<true>(i, 2 * pr, 52)));
  const long pi = ((long)b * n * hw + kp) * 2;
  if (to_tiled) {
    const sv2 v = {(S)plain_in[pi], (S)plain_in[pi + 1]};
    *tp = v;
  } else {
    const sv2 v = *tp;
    plain_out[pi] = (double)v.x;
    plain_out[pi + 1] = (double)v.y;
  }
}

// stand-alone line search support: seed per-problem scalars from caller data
__global__ void k_seed_search(BatchState st, int B, const double *cost, const double *terms) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  st.prev_cost[b] = cost[b];
  st.cost[b] = cost[b];
  st.terms[2 * b] = terms[2 * b];
  st.terms[2 * b + 1] = terms[2 * b + 1];
  st.alpha[b] = 1.0;
  st.trial[b] = 0;
  st.iters[b] = 1;  // so that the Armijo test applies
  st.n_fwd[b] = 0;
  st.status[b] = 0;
  st.flags[b] = F_ACTIVE | F_SEARCH;
}

}  // namespace qilqr

#ifdef QILQR_WITH_SOLVE4  // diagnostics build only (make diag): the one-launch solve, measured behind the rounds
#include "solve4.h"
#endif
