// round_kernels.h -- the combined launches of batches of up to 4 x CUs trajectories: k_backward_rollout (backward pass + rollout) and k_round
// (+ the linearisation of the block's candidates, up to four rounds per launch); bodies: backward4_body.inc, rollout16_body.inc, round_body.inc.
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "backward4_kernel.h"
#include "rollout16_kernel.h"
#include "linearize_kernels.h"

namespace qilqr {

// k_backward_rollout: the two in ONE launch for batches whose blocks of four trajectories all fit the chip at once (one block
// per CU: the rollout's 214 registers): the block's backward pass (fused, barrier-free form), a block barrier, then the
// rollout of its own four trajectories by wavefronts 0..2 -- gains, flags, step sizes written and read by the same CU.  One
// launch boundary and one kernel start fewer per round.
template <typename S>
__global__ __launch_bounds__(320) void k_backward_rollout(ModelConsts<double> c, SolveParams p, BatchState st, int B, int n) {
  {
    constexpr int WAVES = 5;
    constexpr bool FUSED = true, FREE = true, GFAC = false;
    (void)GFAC;
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
// k_round: a whole round in ONE launch, where k_backward_rollout is allowed (a block per CU) -- its backward pass, then its rollout with the
// linearisation of the block's own candidates behind it (round_follow below; until round 6 after it, linearize_block: kept for
// -DQILQR_ROUND_NO_FOLLOW) -- k_linearize's arithmetic: se3_math.h forms its fused
// multiply-adds from the source alone, so the records are the same bits whichever kernel writes them.  One launch boundary and one
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
// The same work BEHIND the rollout (round 6): while three of the block's wavefronts roll the candidates out, the others linearise the knots
// already stored -- tasks (chunk of sixteen knots, half) from a counter in LDS in the order the rollout produces them; lane = (row of the
// block, knot of the chunk) -- and the rollout's wavefronts join when they are through.  With four candidates in a block the round no longer
// ends three passes of lane-tasks after the rollout (22 us) but one task after it (9 us: the last knots' cost half); with one candidate it ends
// as before.  A lane-task's arithmetic is k_linearize's: the same bits.  The step wavefronts announce every stored knot (r16_publish_stores<2>).
template <typename S, int LK>
__device__ __forceinline__ void round_follow(R16Lds &sh, const ModelConsts<S> &c, const S *qr, const BatchState &st, int b, int bs, int buf, bool live,
                                             bool single, int n, int lane) {
  // `single`: ONE of the block's four trajectories rolls out (most launches of a solve's tail).  Its knots are then taken sixty-four to a
  // task AFTER the rollout, as linearize_block does -- a chunk task would have sixteen busy lanes, and followers that work beside a lone
  // rollout cost its chain 2 us per round (their LDS and memory traffic between its hand-offs) for nothing: the round ends one cost half
  // after the rollout either way.  The waiting wavefronts sleep.
  const int per_task = single ? 64 : R16_CHUNK;
  const int groups = (n + per_task - 1) / per_task, ntasks = 2 * groups;
  for (;;) {
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&sh.flags[R16_F_TASK], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= ntasks) break;
    // (the longer chain -- the cost half -- first: of every chunk, and of the whole trajectory when single)
    const bool cost_half = single ? (t < groups) : ((t & 1) == 0);
    const int k0 = (single ? (cost_half ? t : t - groups) : (t >> 1)) * per_task;
    const int need = single ? n : ((k0 + R16_CHUNK < n) ? k0 + R16_CHUNK : n);
    // knots [0, need) are stored when each step wavefront has announced its last knot below `need`
    const int m0 = (need - 1) - ((need - 1) & 1), m1 = (need - 1) - (((need - 1) & 1) ^ 1);
    if (!(r16_flag_wait_relaxed(sh, R16_F_K0, m0 + 1, lane) && (m1 < 0 || r16_flag_wait_relaxed(sh, R16_F_K1, m1 + 1, lane)))) break;
    const int i = k0 + (single ? lane : (lane & 15));
    if (!(single || live) || i >= n) continue;
    const int bb = single ? bs : b;  // (a dead row's lanes carry the first running row's slot and buffer: rollout16_body.inc)
    S pt[18];
    load_knot<true>((const S *)st.traj[buf] + knot_base<true>(bb, n, 18), i, 18, pt);
    S *rec = (S *)st.lin[buf] + rec_base(st.layout, bb, n) + rec_elem(st.layout, i, 0);
    if (!cost_half) {
      const TiledRecWriter<S> wd{rec};
      linearize_dynamics(c, pt, wd);
      wd.flush();
    } else {
      const TiledRecWriter<S> w{rec};
      S pd[18];
      if (st.desired_tiled) load_knot<true>((const S *)st.desired + knot_base<true>(bb, n, 18), i, 18, pd);
      else load_knot<false>((const S *)st.desired, i, 18, pd);
      const S cost = linearize_cost<LK>(qr, qr + 144, pt, pd, w);
      w.flush();
      st.knot_cost[buf][cost_index(bb, i, n)] = (double)cost;
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
// SIX: six wavefronts per block (384 threads) and the backward pass in the six-wavefront form with the matrix wavefronts factoring (round 6).  One
// launch of each k_backward4 form by how many of a block's four trajectories run (profiles/microbench/bw_forms_live.py, us per 100 knots):
//       running     fused     six wavefronts, knot loop unrolled
//          1        67.3            61.5
//          2        67.2            62.3
//          3        67.7            69.6
//          4        67.4            72.1
// -- in the fused form the gradient recursion rides on the matrix wavefront's own instruction stream whatever the block holds; in the
// six-wavefront form it is another wavefront's, and the per-knot barrier costs by how many matrix wavefronts meet at it.  The bits are the
// same (round 6), so the host takes this form for the launches in which, on average, at most two trajectories per block still run: the late
// rounds, where the slowest problem of a batch is alone in its block.  Inside k_round most of the stand-alone kernels' 5.7 us does not
// arrive: a lone trajectory's round takes 126.0 us instead of 127.9 (B = 1, four rounds per launch; with ONE round per launch the form is
// 3.4 us per round SLOWER: something per launch of the 384-thread block), a B = 1024 solve 4.739 instead of 4.756 ms on configs[1]'s
// problems (- 0.4 %) and 5.554 instead of 5.646 on configs[3]'s (- 1.6 %), B = 64 - 0.6 %; forced in every launch it loses 2.6 % at
// B = 1024 (the early rounds, four running trajectories per block).  Kept: never slower where the host takes it.
template <int LK, int ROUNDS, bool SIX = false>
__global__ __launch_bounds__(SIX ? 384 : 320) void k_round(ModelConsts<double> c, const ModelConsts<double> *__restrict__ cp, SolveParams p, BatchState st, int B,
                                                int n, int *prev_counters, int prev_round) {
  typedef double S;
  __shared__ double qr_w[160];  // the weights of the cost half (k_linearize keeps a copy per wavefront: here the block's)
  BW4_DECLARE_LDS
  __shared__ R16Lds sh;
#ifdef QILQR_ROUND_STAMPS
  __shared__ unsigned long long round_stamp_x0;
  unsigned long long rs_t0 = 0, rs_t1 = 0, rs_t2 = 0, rs_t3 = 0;
#endif
#define BW4_LDS_DECLARED
#define R16_LDS_DECLARED
#define BW4_CTAB_FILLED
  for (int k = threadIdx.x; k < 160; k += blockDim.x) qr_w[k] = (k < 144) ? cp->Q[k] : cp->R[k - 144];
  if (threadIdx.x < 64) r16_exp2_lds[threadIdx.x >> 4][threadIdx.x & 15] = Series<double>::exp2[threadIdx.x >> 4][threadIdx.x & 15];  // (rollout16_body.inc)
  // the constant operand table behind the ring slots, once for all the rounds of the launch (a round whose block has nothing to run
  // leaves before it would fill it, and a later round of the same launch may have something: so here, unconditionally; the records and
  // the settle step's scratch use the slots' other words)
  bw4_fill_ctab<S>(ring, st.ctab, SIX ? 384 : 320);
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
  // (More than four rounds per launch, measured in round 6 at B = 1024 -- a launch boundary costs 5-10 us, 4.715 / 4.628 / 4.545 ms per solve
  // with 1 / 2 / 4 rounds per launch: EIGHT copies 5.25 ms, and a LOOP over one copy, for 8, 16 or 32 rounds, 6.4 ms: in a loop the compiler
  // keeps the round's invariants in registers across the iterations and the kernel spills 1 KB per lane.  Four copies stay.)
#undef ROUND_BEHIND_BACKWARD
#undef BW4_LDS_DECLARED
#undef R16_LDS_DECLARED
#undef BW4_CTAB_FILLED
}

}  // namespace qilqr
