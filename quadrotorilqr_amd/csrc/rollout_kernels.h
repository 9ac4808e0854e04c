// rollout_kernels.h -- k_rollout (a lane per trajectory, one wavefront: the Runge-Kutta extension's kernel) and k_rollout3 (a lane per trajectory, pose /
// control / loader wavefronts: batches beyond 4096 trajectories) -- ilqr.hh:149-172.
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "kernels_common.h"

namespace qilqr {

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

}  // namespace qilqr
