// backward4_kernel.h -- k_backward4: the backward pass over FOUR trajectories per block -- the roles (gradient, loader, matrix, fused matrix + gradient
// wavefronts), the block's LDS, the kernel; its body is backward4_body.inc (k_backward_rollout and k_round contain it too).
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "backward_common.h"

namespace qilqr {

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
// The two sums of the gradient recursion whose ORDER is part of the arithmetic, written with explicit fused multiply-adds (until round 6 they
// were `a*b + c*d + ...` expressions and the order was the compiler's choice of contraction -- m[1] vxl[1] first, as it happened; the six-
// wavefront form performs the same operations, bw4f_gradient_wave, and a backward pass no longer depends on the batch size).
__device__ __forceinline__ double bw4_quarter_sum(const double (&m)[3], const double (&v)[3]) {  // rows kk, 4 + kk, 8 + kk of M^T V_x
  return __builtin_fma(m[2], v[2], __builtin_fma(m[0], v[0], m[1] * v[1]));
}
__device__ __forceinline__ double bw4_chain4(const double (&a)[4], const double (&b)[4]) {  // K[:, j]^T Q_u; in lane 12: Q_u^T k
  return __builtin_fma(a[3], b[3], __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], a[0] * b[0])));
}
// acc += m * (vx of lane R of the caller's row of 16): one v_fmac_f64_dpp (the compiler keeps broadcast and
// multiply-add apart).  vx must have been written at least two instructions earlier (DPP read hazard): it is
// the previous knot's result here.
template <int R>
__device__ __forceinline__ double bw4_dot_step(double acc, double m, double vx) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vx), "v"(m), "n"(R));
  return acc;
}
// m * (vx of lane R of the caller's row): the first term of a chain.  gfx950 has no DPP form of v_mul_f64; a multiply-add onto -0.0 is the
// product to the last bit, the sign of a zero product included (x + -0.0 = x for every x, and -0.0 + -0.0 = -0.0).
template <int R>
__device__ __forceinline__ double bw4_mul_step(double m, double vx) { return bw4_dot_step<R>(-0.0, m, vx); }
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
  // (the gradient wavefront holds its block's matrix wavefronts at every knot's barrier: it issues at their priority, as in the form below --
  // round 6; within noise alone, + 0.5-1 % where several sub-batches' blocks share a CU)
  __builtin_amdgcn_s_setprio(3);
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
    // [Q_x ; Q_u][j] = [C_x ; C_u][j] + sum_r M[r][j] V_x[r] in the FUSED form's order (round 6: a backward pass's bits do not depend on
    // the batch size): the quarter kk sums its three rows as one lane of bw4_fused_wave does -- the product of row 4 + kk, then
    // multiply-adds of rows kk and 8 + kk (bw4_quarter_sum) --, then its two butterflies: (p0 + p1) + (p2 + p3).  (Until round 5: three
    // chains of four rows; the fused order had been tried in round 3 and dropped because the forms still differed elsewhere -- in the
    // order H is accumulated, as it turned out.)
    double p0 = bw4_mul_step<4>(m[4], vx), p1 = bw4_mul_step<5>(m[5], vx), p2 = bw4_mul_step<6>(m[6], vx), p3 = bw4_mul_step<7>(m[7], vx);
    p0 = bw4_dot_step<0>(p0, m[0], vx); p1 = bw4_dot_step<1>(p1, m[1], vx); p2 = bw4_dot_step<2>(p2, m[2], vx); p3 = bw4_dot_step<3>(p3, m[3], vx);
    p0 = bw4_dot_step<8>(p0, m[8], vx); p1 = bw4_dot_step<9>(p1, m[9], vx); p2 = bw4_dot_step<10>(p2, m[10], vx); p3 = bw4_dot_step<11>(p3, m[11], vx);
    const double ghat = gcj + ((p0 + p1) + (p2 + p3));
    const double Qu0 = row_bcast<12>(ghat), Qu1 = row_bcast<13>(ghat), Qu2 = row_bcast<14>(ghat),
                 Qu3 = row_bcast<15>(ghat);
    // V_x = Q_x + K^T Q_u, the sum as the fused form's chain (bw4_chain4): the recurrence ends here
    vx = ghat + __builtin_fma(c3, Qu3, __builtin_fma(c2, Qu2, __builtin_fma(c1, Qu1, c0 * Qu0)));
    double kff[4];
    ldlt4_solve_neg(Ldlt4{l10, l20, l30, l21, l31, l32, i0, i1, i2, i3}, Qu0, Qu1, Qu2, Qu3, kff);
    const double k0 = kff[0], k1 = kff[1], k2 = kff[2], k3 = kff[3];  // feed-forward (ilqr.hh:128)
    const sv2 w0 = {(S)k0, (S)k1}, w1 = {(S)k2, (S)k3};
    *kdst0 = w0;
    *kdst1 = w1;
    kdst0 -= kst;
    kdst1 -= kst;
    QuTk += __builtin_fma(k3, Qu3, __builtin_fma(k2, Qu2, __builtin_fma(k1, Qu1, k0 * Qu0)));  // (lane 12's sum in the fused form)
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
template <typename S, bool FREE = false, bool TWO = false>  // TWO: two block barriers per knot (the bw4f_* roles)
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
    if constexpr (TWO) __syncthreads();
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
    d4 H = bw_tile_H201(mc, T, cc, cuu);
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
      d4 H = bw_tile_H201(m, T, cx, cuu);
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
    const double ktq = bw4_chain4(kp, Qup);
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
    double part = bw4_quarter_sum(m, vxl);  // [Q_x ; Q_u] = [C_x ; C_u] + M^T V_x
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
    QuTk += bw4_chain4(kp, Qup);
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
  if (dead) bw4_prog_post(prog, 5, 1, lane);
  return bcast_lane(QuTk, 12);
}

// ---- Round 6: the six-wavefront form with the FACTORISATION IN THE GRADIENT WAVEFRONT (bw4f_*; what k_backward4<S, WAVES, false>
// runs; the roles above stay for the diagnostics build's k_solve4).
// In the form above every one of a matrix wavefront's 64 lanes computes the same LDL^T of Q_uu -- 37 of its 52 fp64 vector instructions
// and ten broadcasts per knot, four times per block -- and beyond 4096 trajectories, with four blocks on a CU, a SIMD is bound by what
// its wavefronts ISSUE.  Q_uu = C_uu + J_u^T V_xx J_u needs nothing of the knot but the previous knot's V_xx, and of that only the
// 4 x 4 block of rows and columns 8..11 (J_u is zero above row 8): the matrix wavefront leaves that block in LDS behind its V_xx product
// (kw), the gradient wavefront -- a row of 16 lanes per trajectory, lane 4 a + b of a row for Q_uu[a][b] -- forms Q_uu by 20 multiply-adds,
// factors all four trajectories' in one instruction stream and hands ten numbers back (behind K in kf), while the matrix wavefronts are in their
// T and H products; they wait for it at a second block barrier per knot, in front of the substitution.
// SAME BITS as the accumulator tile's rows 12..15: v_mfma_f64_16x16x4_f64 is, per element, the chain of four fused multiply-adds
// k = 0..3 on the accumulator (profiles/microbench/mfma_arith.hip: 0 of 2^20 elements differ, against 45 % for any other order), the
// products with J_u's zero rows add exact zeros, so Q_uu[a][b] = fma-chain_r(J_u[8+r][a], T[8+r][12+b]; C_uu[a][b]) with
// T[8+r][12+b] = fma-chain_s(V[8+s][8+r], J_u[8+s][b]; 0) is what the tile holds, to the last bit (a V_xx with infinities aside: 0 x inf
// in the rows the chain skips).  And the same bits as the FUSED form (bw4_fused_wave, up to 4096 trajectories): H is accumulated in
// its order (kc = 2, 0, 1), M^T V_x as its three multiply-adds per quarter and (p0 + p1) + (p2 + p3), K^T Q_u as its chain -- a
// backward pass no longer depends on the batch size (tests/test_gpu_parity.py::test_backward_pass_bits_do_not_depend_on_the_batch_size).
// Every role executes exactly 1 + 2 n block barriers: interval i = [A_i: W(i+1) and K(i+1) are in LDS | G factors Q_uu(i); M forms T, H,
// gathers | B_i: the factors are in LDS | M substitutes, stores, V_xx(i), W(i); G's gradient step of knot i+1 | A_(i-1)].
template <typename S>
__device__ __forceinline__ double bw4f_gradient_wave(double (&ring)[4][4][BW2_BUF], double (&kf)[4][2][80], double (&kw)[4][16],
                                                     const RecLayout &L, S *gains, S *dump4, bool grun,
                                                     double cuu_ab, int n, int lane) {
  typedef typename GA<S>::v2 sv2;
  typedef typename GA<S>::ptr2 gptr2;
  const int g = lane >> 4, j = lane & 15;
  const double *mp[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const int src = m_source_tab(r, j);
    mp[r] = &ring[g][0][(src >= 0) ? src : BW2_REC + (-1 - src)];
  }
  const double *gp = &ring[g][0][L.off_g + j];
  const bool kowner = grun && (j == 0);
  // (one pointer, the second pair's distance and the knot stride as 32-bit lane values: three registers fewer than two pointers and a
  // 64-bit stride, in a wavefront that has 80)
  gptr2 kdst = (gptr2)(kowner ? gains + knot_elem<true>(n - 1, 0, 52) : dump4);
  const int kd1 = kowner ? (int)((knot_elem<true>(0, 2, 52) - knot_elem<true>(0, 0, 52)) / 2) : 1;
  const int kst = kowner ? (int)((knot_elem<true>(1, 0, 52) - knot_elem<true>(0, 0, 52)) / 2) : 0;
  double vx = 0.0;  // V_x[j] (lanes j < 12)
  double QuTk = 0.0;
  __syncthreads();                               // rings, constant tables, W = 0
  // J_u's rows 8..11 as this lane needs them: column b for T, column a for Q_uu (the constant table behind ring slot 0; read again at
  // every knot -- eight registers that stay are eight the gradient step does not have: 80 per lane at four blocks per CU)
  const int qa = j >> 2, qb = j & 3;
  const double *bpb = &ring[g][0][BW2_REC + CTAB_BU + 8 * 4 + qb], *bpa = &ring[g][0][BW2_REC + CTAB_BU + 8 * 4 + qa];
  const double *wq = &kw[g][0];
  // the factors go where the matrix wavefront used to leave them: behind K in the hand-off buffer of the knot's parity
  auto factor_step = [&](int par) {
    double wv[16], bub[4], bua[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) wv[e] = wq[e];   // [4 r + s] = V[8 + s][8 + r], the row's sixteen lanes read the same words
#pragma unroll
    for (int r = 0; r < 4; ++r) { bub[r] = bpb[4 * r]; bua[r] = bpa[4 * r]; }
    double q = cuu_ab;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double t = 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s) t = __builtin_fma(wv[4 * r + s], bub[s], t);  // T[8 + r][12 + b]: the kc = 2 product of T = V M
      q = __builtin_fma(bua[r], t, q);                                         // H[12 + a][12 + b]: the kc = 2 product of H
    }
    double Quu[16];
    Quu[0] = row_bcast<0>(q);
    Quu[4] = row_bcast<4>(q); Quu[5] = row_bcast<5>(q);
    Quu[8] = row_bcast<8>(q); Quu[9] = row_bcast<9>(q); Quu[10] = row_bcast<10>(q);
    Quu[12] = row_bcast<12>(q); Quu[13] = row_bcast<13>(q); Quu[14] = row_bcast<14>(q); Quu[15] = row_bcast<15>(q);
    Quu[1] = Quu[2] = Quu[3] = Quu[6] = Quu[7] = Quu[11] = 0.0;
    const Ldlt4 f = ldlt4_factor(Quu);  // (ilqr.hh:126; unpivoted: see ldlt4_factor)
    if (j == 0) {
      typedef double dv2 __attribute__((ext_vector_type(2)));
      dv2 *o = reinterpret_cast<dv2 *>(&kf[g][par][64]);
      o[0] = dv2{f.l10, f.l20}; o[1] = dv2{f.l30, f.l21}; o[2] = dv2{f.l31, f.l32}; o[3] = dv2{f.i0, f.i1}; o[4] = dv2{f.i2, f.i3};
    }
  };
  auto gradient_slot = [&](int q, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const double *f = kf[g][q & 1];
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) m[r] = mp[r][SLOT * BW2_BUF];
    const double gcj = gp[SLOT * BW2_BUF];
    asm volatile("" ::: "memory");
    const double c0 = f[4 * j], c1 = f[4 * j + 1], c2 = f[4 * j + 2], c3 = f[4 * j + 3];  // K[:, j]
    const Ldlt4 f4 = {f[64], f[65], f[66], f[67], f[68], f[69], f[70], f[71], f[72], f[73]};  // (this wavefront's own, of one interval ago)
    asm volatile("" ::: "memory");
    // [Q_x ; Q_u][j] = [C_x ; C_u][j] + sum_r M[r][j] V_x[r] in the fused form's order: the quarter kk sums its three rows as one lane of
    // bw4_fused_wave does -- the product of row 4 + kk, then multiply-adds of rows kk and 8 + kk (bw4_quarter_sum) --, then its two
    // butterflies: (p0 + p1) + (p2 + p3)
    double p0 = bw4_mul_step<4>(m[4], vx), p1 = bw4_mul_step<5>(m[5], vx), p2 = bw4_mul_step<6>(m[6], vx), p3 = bw4_mul_step<7>(m[7], vx);
    p0 = bw4_dot_step<0>(p0, m[0], vx); p1 = bw4_dot_step<1>(p1, m[1], vx); p2 = bw4_dot_step<2>(p2, m[2], vx); p3 = bw4_dot_step<3>(p3, m[3], vx);
    p0 = bw4_dot_step<8>(p0, m[8], vx); p1 = bw4_dot_step<9>(p1, m[9], vx); p2 = bw4_dot_step<10>(p2, m[10], vx); p3 = bw4_dot_step<11>(p3, m[11], vx);
    const double ghat = gcj + ((p0 + p1) + (p2 + p3));
    const double Qu0 = row_bcast<12>(ghat), Qu1 = row_bcast<13>(ghat), Qu2 = row_bcast<14>(ghat),
                 Qu3 = row_bcast<15>(ghat);
    // V_x = Q_x + K^T Q_u, the sum as the fused form's chain: k0 Q0, then three multiply-adds
    const double ktq = __builtin_fma(c3, Qu3, __builtin_fma(c2, Qu2, __builtin_fma(c1, Qu1, c0 * Qu0)));
    vx = ghat + ktq;  // the recurrence ends here
    double kff[4];
    ldlt4_solve_neg(f4, Qu0, Qu1, Qu2, Qu3, kff);
    const double k0 = kff[0], k1 = kff[1], k2 = kff[2], k3 = kff[3];  // feed-forward (ilqr.hh:128)
    const sv2 w0 = {(S)k0, (S)k1}, w1 = {(S)k2, (S)k3};
    kdst[0] = w0;
    kdst[kd1] = w1;
    kdst -= kst;
    QuTk += __builtin_fma(k3, Qu3, __builtin_fma(k2, Qu2, __builtin_fma(k1, Qu1, k0 * Qu0)));  // (lane 12's sum in the fused form)
  };
  auto gradient_step = [&](int q) {
    switch (q & 3) {
      case 0: gradient_slot(q, std::integral_constant<int, 0>()); break;
      case 1: gradient_slot(q, std::integral_constant<int, 1>()); break;
      case 2: gradient_slot(q, std::integral_constant<int, 2>()); break;
      default: gradient_slot(q, std::integral_constant<int, 3>()); break;
    }
  };
  // The gradient wavefront is on its block's critical path twice per knot (the matrix wavefronts wait for its factors, then for its gradient
  // step): it issues at the matrix wavefronts' priority.  At the default priority it waits behind the matrix wavefronts of the OTHER blocks
  // on its SIMD while its own block's sit at the barrier: 3615 against 3515 us per launch at B = 65536, 520 / 484 at 8192.
  __builtin_amdgcn_s_setprio(3);
  for (int i = n - 1; i >= 0; --i) {
    factor_step(i & 1);   // Q_uu(i) from W(i + 1)
    __syncthreads();      // B_i: the matrix wavefronts take the factors
    if (i + 1 <= n - 1) gradient_step(i + 1);
    __syncthreads();      // A_(i-1)
  }
  gradient_step(0);
  return QuTk;
}
// M_w in that form: T, H (accumulated in the fused form's order), the row gather; the factors from G; the substitution, the gain stores, K
// for G, V_xx, and the block of V_xx that the next Q_uu is made of.
template <typename S, bool UNROLL = false>
__device__ __forceinline__ void bw4f_matrix_wave(double (&ring)[4][4][BW2_BUF], double (&kf)[4][2][80], double (&kw)[4][16],
                                                 const RecLayout &L, int w, bool run, S *gains, S *dump4, double cuu,
                                                 int n, int lane, unsigned long long *stamps_out) {
  typedef typename GA<S>::v2 sv2;
  typedef typename GA<S>::ptr2 gptr2;
  typedef double dv2 __attribute__((ext_vector_type(2)));
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
  // W = rows / columns 8..11 of the accumulator tile: lane (j = 8 + r, kk = s) holds V[8 + s][8 + r] in register 2 -> kw[w][4 r + s]
  const bool wowner = (j >= 8 && j < 12);
  double *const wdst = &kw[w][wowner ? 4 * (j - 8) + kk : 0];
  if (wowner) *wdst = 0.0;         // V_xx = 0 behind the last knot: Q_uu(n - 1) = C_uu
  __syncthreads();                 // rings and constant tables are filled
  if (!run) {
    for (int i = n - 1; i >= 0; --i) { __syncthreads(); __syncthreads(); }
    return;
  }
  double m[3], cx[3];
  {
    const double *buf = ring[w][(n - 1) & 3];
    m[0] = buf[off[0]]; m[1] = buf[off[1]]; m[2] = buf[off[2]];
    cx[0] = buf[off[3]]; cx[1] = buf[off[4]]; cx[2] = buf[off[5]];
  }
  asm volatile(".p2align 6");
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev, real0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  const double *rp[6];
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
    d4 H = bw_tile_H201(mc, T, cc, cuu);  // (the fused form's order: part of the arithmetic)
    QKEEP(H[0]); QKEEP(H[3]);
    QSTAMP(1);  // H = C + M^T T
    double col[4];
    gather_rows(H[3], col);
    QKEEP(col[0]); QKEEP(col[3]);
    QSTAMP(2);  // row gather
    double *f = kfw + (int)par * 80;
    __syncthreads();  // B: G has left the factors of this knot's Q_uu behind K's place in the hand-off buffer
    const dv2 *fq = reinterpret_cast<const dv2 *>(f + 64);
    const dv2 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3], f4v = fq[4];
    const Ldlt4 f4 = {f0.x, f0.y, f1.x, f1.y, f2.x, f2.y, f3.x, f3.y, f4v.x, f4v.y};
    QSTAMP(3);  // barrier, factors
    double kcol[4];
    ldlt4_solve_neg(f4, col[0], col[1], col[2], col[3], kcol);  // K[:, j] (ilqr.hh:127)
    QKEEP(kcol[0]); QKEEP(kcol[3]);
    QSTAMP(4);  // substitution
    {
      const sv2 w0 = {(S)kcol[0], (S)kcol[1]}, w1 = {(S)kcol[2], (S)kcol[3]};
      *gdst0 = w0;
      *gdst1 = w1;
      gdst0 -= gstep;
      gdst1 -= gstep;
    }
    // hand K to G (the four lanes of a column hold the same K[:, j]: same address, same data)
    f[4 * j] = kcol[0]; f[4 * j + 1] = kcol[1]; f[4 * j + 2] = kcol[2]; f[4 * j + 3] = kcol[3];
    QSTAMP(5);  // gain stores, hand-off to G
    // V_xx = Q_xx + Q_xu K: A[j][kk] = Q_xu[j][kk] = H[12 + kk][j] is accumulator register 3
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(H[3], sel4(kcol, kk), H, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) va[kc] = H[kc];
    if (wowner) *wdst = H[2];
    QKEEP(va[0]); QKEEP(mn[2]);
    QSTAMP(6);  // V_xx MFMA, W, next operands
    __syncthreads();  // A
    QSTAMP(7);  // barrier
  };
  double mn[3], cn[3];
  int i = n - 1;
  if constexpr (UNROLL) {
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
    for (; i >= 0; --i) {
      knot((i > 0 ? i - 1 : 0) & 3, i & 1, m, cx, mn, cn);
#pragma unroll
      for (int k = 0; k < 3; ++k) { m[k] = mn[k]; cx[k] = cn[k]; }
    }
  }
#ifdef QILQR_STAMPS
  {
    unsigned long long real1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");
    stamp_sum[3] = (real1 - real0) & 0xfffffull;
  }
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
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
  __shared__ __attribute__((aligned(16))) double kf[4][2][80];           \
  __shared__ __attribute__((aligned(16))) double kw[4][16];              \
  __shared__ double s_mu[4];                                             \
  __shared__ int prog[24];
#define QILQR_CAT_(a, b) a##b
#define QILQR_CAT(a, b) QILQR_CAT_(a, b)
// GFAC (six wavefronts only): the factorisation of Q_uu in the gradient wavefront (bw4f_* roles, two block barriers per knot) -- the form for
// a saturated chip; without it the matrix wavefronts factor (one barrier per knot) -- the form for launches whose wavefronts are alone
// on their SIMDs.  THE SAME BITS either way: the host picks by how many trajectories are running (launch_backward).
template <typename S, int WAVES, bool FUSED = false, bool FREE = false, bool GFAC = false>
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

}  // namespace qilqr
