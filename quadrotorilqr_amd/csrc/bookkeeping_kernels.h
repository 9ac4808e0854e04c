// bookkeeping_kernels.h -- the kernels around the rounds: k_accept (stand-alone line search, ilqr.hh:174-194), k_gather / k_retile (layout conversion at
// the ABI), the compaction of the running trajectories (k_compact_plan, k_compact_move), the debug ring of the single solve
// (k_debug_capture), the copy-back under the tail (k_mark_final, k_late_slots), k_seed_search.
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "kernels_common.h"

namespace qilqr {

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
// The same by ONE wavefront inside k_round (round 6): the block's fifth wavefront -- the backward pass's loader, idle while wavefronts 0..2
// roll out -- looks behind every round's backward pass, so that a single solve with populate_debug (the reference's default,
// quadrotor_ilqr.py:283) keeps four rounds per launch instead of one launch + k_debug_capture per round.  The settle step's stores (iters,
// cur, cost: one lane of matrix wavefront 0) are in front of the block barrier this runs behind; the rollout beside it reads the current
// trajectory and writes the other buffer.
template <typename S>
__device__ __forceinline__ void debug_capture_wave(const BatchState &st, int n, int lane) {
  const int seen = __builtin_amdgcn_readfirstlane(*st.dbg_seen), it = __builtin_amdgcn_readfirstlane(st.iters[0]);
  if (it <= seen) return;
  if (seen < st.dbg_cap) {
    typedef typename GA<S>::v2 sv2;
    const S *t = (const S *)st.traj[st.cur[0]] + knot_base<true>(0, n, 18);
    double *o = st.dbg_trajs ? st.dbg_trajs + (size_t)seen * n * 18 : nullptr;
    if (o)
      for (int kp = lane; kp < n * 9; kp += 64) {
        const int i = kp / 9, pr = kp - 9 * i;
        const sv2 v = *reinterpret_cast<const sv2 *>(t + knot_elem<true>(i, 2 * pr, 18));
        o[2 * kp] = (double)v.x;
        o[2 * kp + 1] = (double)v.y;
      }
    if (lane == 0 && st.dbg_cost) st.dbg_cost[seen] = st.cost[0];
  }
  if (lane == 0) *st.dbg_seen = it;
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
