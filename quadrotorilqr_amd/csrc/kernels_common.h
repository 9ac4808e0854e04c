// kernels_common.h -- what every kernel shares: the workspace (BatchState), solve parameters, section stamps of the diagnostics builds,
// the block -> XCD map, the convergence test and the stores of the per-trajectory state machine (store_settled, arm_line_search).
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
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
  // ILQRDebug of the single solve (ilqr.hh:78-80): the ring the accepted trajectories and their costs go to (null outside qilqr_solve with
  // populate_debug): written by k_debug_capture behind a round of separate launches, or by an idle wavefront of k_round inside the launch
  double *dbg_trajs, *dbg_cost;
  int *dbg_seen;
  int dbg_cap;
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

}  // namespace qilqr
