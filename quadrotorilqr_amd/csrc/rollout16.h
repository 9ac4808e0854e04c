// rollout16.h -- the closed-loop forward simulation (ILQR::forward_sim, ilqr.hh:149-172) with SIXTEEN LANES PER
// TRAJECTORY: one wavefront rolls out four trajectories, one per row of 16 lanes.
//
// Why.  With a lane per trajectory (k_rollout3) a knot is ~600 fp64 instructions spread over two cooperating
// wavefronts, and a lone wavefront issues one fp64 instruction per ~6.8 cycles whatever the exec mask: ~2000
// cycles per knot on 16 of the chip's 256 CUs at B = 1024.  Spending the idle lanes on the INSIDE of a trajectory
// cuts the instruction count of the knot to about a third on ONE wavefront:
//   * every product with a matrix that does not depend on the state -- the nominal rotation R_n^T, the left
//     multiplication by conj(q_n), the 4 x 12 feedback gain, I, dt I^-1, dt I^-1 arms -- is one
//     v_fmac_f64_dpp row_newbcast per INPUT component (the lane holds its row of the matrix, the broadcast brings
//     the component): 4 x 12 gains = 12 instructions instead of 52, a nominal quaternion product 4 instead of 28;
//   * the seven polynomials of the knot (cos, sinc, two Jacobian coefficients of Exp; the two halves of asin for
//     Log) are evaluated side by side in different lanes with per-lane coefficient registers (one Estrin pass);
//   * cross products are done four at a time: three-vectors sit in lanes 0..2 of a quad, a x b = a' b'' - a'' b'
//     with ' a rotation inside every quad (two v_mov_b32_dpp quad_perm per 64-bit value -- fp64 DPP itself has
//     only row_newbcast on gfx950), so th x td (Log), theta_e x rho_e (Exp) and omega x I omega (gyroscopic term)
//     share one set of rotations;
//   * elementwise vector operations are one instruction instead of three to six.
// Everything that depends only on the nominal trajectory and the gains is prepared by a second wavefront P, one
// knot ring ahead, as 23 "operand registers" per knot in LDS; the rollout wavefront R reads its lane's 23 values.
//
// The arithmetic is the reference's (se3_math.h: se3_rminus_fast / se3_rplus_fast / control_law /
// body_acceleration_fast) with other summation orders: results agree with the oracle to rounding (1e-12).
//
// The code is written once against a "wave" policy W (value-per-lane type V, predicate M, cross-lane operations):
// DevWave maps it to registers and DPP on gfx950; tests/host_harness.cpp maps it to arrays of 64 lanes so that the
// CPU suite checks the lane maps, the operand preparation and the arithmetic against the oracle without a GPU.
//
// Lane map of a row (16 lanes = quads Q0..Q3, j = lane & 3):
//   three-vectors live in lanes j = 0..2 of a quad; the quaternion (x, y, z, w) in j = 0..3 of every quad
//   TT  translation in Q1          QQ  quaternion in every quad        VL, VW  linear / angular velocity in every quad
//   Log works in Q0 (td, theta, rho), Exp in Q1 (theta_e, rho_e, p), the gyroscopic term in Q2, the control in Q3.
#pragma once
#include "se3_math.h"

// branches a converging solve does not take: laid out off the knot loop's straight line (a taken branch costs a lone
// wavefront an instruction refetch)
#define QILQR_RARE(x) __builtin_expect(!!(x), 0)

namespace qilqr {
namespace r16 {

// ---------------------------------------------------------------------------------------------------------------
// operand registers prepared by P for one knot (value per lane; rows of matrices live in the lanes that consume them)
// Every operand is used through one quad of a row only -- as the multiplier of a broadcast-multiply-add whose result is read in
// that quad, or as a term of a difference that is then broadcast from that quad -- so operands that live in different quads
// SHARE a register: a step wave reads 13 registers per knot from LDS, not 23 (the reads' issue was a quarter of a step).
// What a register holds in the quads its users do not look at is data of another operand: results there are finite and unread.
enum {
  OP_K = 0,     // 12: Q3: column c of the feedback gain K (4 x 12) as rows;
                //     Q0 of registers 0..2: column c of R_n^T as rows (lane (Q0, j) holds R_n[c][j]) = OP_RT + c,
                //     Q0 of registers 3..6: column c of the left-multiplication matrix of conj(q_n), rows (x, y, z, w) = OP_LQ + c
  OP_RT = 0,
  OP_LQ = 3,
  OP_MISC = 12, // Q0: nominal linear velocity, Q1: nominal translation, Q2: nominal angular velocity (j < 3 each), Q3: u_nom + alpha k
  OP_TN = OP_MISC, OP_U0 = OP_MISC, OP_VNL = OP_MISC, OP_VNW = OP_MISC,
  NOPS = 13
};

// ---------------------------------------------------------------------------------------------------------------
// lane constants of the rollout wavefront (built once per launch)
template <class W>
struct RConsts {
  typedef typename W::V V;
  typedef typename W::M M;
  double dt;
  V SA, SB1;           // scales: SA VW = [0 | theta_e | omega | 0], SB1 VL = [0 | rho_e | 0 | 0]
  V IC[3];             // column c of the inertia, rows in Q2
  V NI[3];             // column c of -dt I^-1, rows in every quad
  V GLz;               // dt / m at j = 2
  V GW[4];             // column a of dt I^-1 arms, rows in every quad
  V GRAVC, S1, S2;     // gravity: -g dt e_z, and the signed scales of the two quaternion permutations
  V MQ1, MQ1_3, MW;
  V MQ0_3, MQ2_3, MQ3, MQ0; // store assembly
  V M3;                // 1 in lanes j < 3
  V PC[8];             // per-lane polynomial coefficients (lane-in-row 0..5)
  M L0, L1, L2, L3;    // lane-in-row == 0..3 (patching the Exp coefficients on the closed-form path)
};

template <class W>
QILQR_HD void make_rconsts(const ModelConsts<double> &c, RConsts<W> &k) {
  const double dt = c.dt;
  k.dt = dt;
  auto q = [](int l) { return (l >> 2) & 3; };
  auto j = [](int l) { return l & 3; };
  k.SA = W::vconst([&](int l) { return (j(l) < 3) ? (q(l) == 1 ? dt : (q(l) == 2 ? 1.0 : 0.0)) : 0.0; });
  k.SB1 = W::vconst([&](int l) { return (j(l) < 3 && q(l) == 1) ? dt : 0.0; });
  for (int cc = 0; cc < 3; ++cc) {
    k.IC[cc] = W::vconst([&](int l) { return (j(l) < 3 && q(l) == 2) ? c.inertia[3 * j(l) + cc] : 0.0; });
    k.NI[cc] = W::vconst([&](int l) { return (j(l) < 3) ? -dt * c.inertia_inv[3 * j(l) + cc] : 0.0; });
  }
  k.GLz = W::vconst([&](int l) { return j(l) == 2 ? c.Bu[8 * 4] : 0.0; });  // row 8 of J_u: dt / m (the same for every rotor)
  for (int a = 0; a < 4; ++a) k.GW[a] = W::vconst([&](int l) { return (j(l) < 3) ? c.Bu[(9 + j(l)) * 4 + a] : 0.0; });
  k.GRAVC = W::vconst([&](int l) { return j(l) == 2 ? -c.g * dt : 0.0; });
  // R^T e_z = (2(xz - yw), 2(yz + xw), 1 - 2(x^2 + y^2)) = x (2z, 2w, -2x) + y (-2w, 2z, -2y) + e_z
  k.S1 = W::vconst([&](int l) { return -c.g * dt * (j(l) == 0 ? 2.0 : (j(l) == 1 ? 2.0 : (j(l) == 2 ? -2.0 : 0.0))); });
  k.S2 = W::vconst([&](int l) { return -c.g * dt * (j(l) == 0 ? -2.0 : (j(l) == 1 ? 2.0 : (j(l) == 2 ? -2.0 : 0.0))); });
  k.MQ1 = W::vconst([&](int l) { return q(l) == 1 ? 1.0 : 0.0; });
  k.MQ1_3 = W::vconst([&](int l) { return (q(l) == 1 && j(l) < 3) ? 1.0 : 0.0; });
  k.MW = W::vconst([&](int l) { return j(l) == 3 ? 1.0 : 0.0; });
  k.MQ0_3 = W::vconst([&](int l) { return (q(l) == 0 && j(l) < 3) ? 1.0 : 0.0; });
  k.MQ2_3 = W::vconst([&](int l) { return (q(l) == 2 && j(l) < 3) ? 1.0 : 0.0; });
  k.MQ3 = W::vconst([&](int l) { return q(l) == 3 ? 1.0 : 0.0; });
  k.MQ0 = W::vconst([&](int l) { return q(l) == 0 ? 1.0 : 0.0; });
  k.M3 = W::vconst([&](int l) { return j(l) < 3 ? 1.0 : 0.0; });
  for (int e = 0; e < 8; ++e)
    k.PC[e] = W::vconst([&](int l) {
      switch (l & 15) {
        case 0: return Series<double>::cos_half[e];
        case 1: return Series<double>::sin_half_over[e];
        case 2: return Series<double>::jac_a[e];
        case 3: return Series<double>::jac_b[e];
        case 4: return 2.0 * Series<double>::asin_lo[e];   // Log's coefficient is 2 asin(s) / s
        case 5: return 2.0 * Series<double>::asin_hi[e];
        default: return 0.0;
      }
    });
  k.L0 = W::mconst([](int l) { return (l & 15) == 0; });
  k.L1 = W::mconst([](int l) { return (l & 15) == 1; });
  k.L2 = W::mconst([](int l) { return (l & 15) == 2; });
  k.L3 = W::mconst([](int l) { return (l & 15) == 3; });
}

// a x b for three-vectors in lanes j = 0..2 of every quad, given the rotations a' (j <- j + 1) and a'' (j <- j + 2) of a;
// lane j = 3 of every quad comes out as a[3] b[3] - a[3] b[3] = 0
template <class W>
QILQR_HD typename W::V cross_rot(typename W::V ap, typename W::V app, typename W::V b) {
  const typename W::V bp = W::rot1(b), bpp = W::rot1(bp);
  return W::fma(ap, bpp, -(app * bp));
}

// One step of the rollout for four trajectories, as the pieces the device code composes (rollout16_kernel.h, r16_wave_X; the
// host harness composes them the same way).  The recurrence has a two-knot period -- the pose of knot i + 2 needs v_{i+1},
// which needs u_i, which needs the pose of knot i:  Log_i -> u_i -> v_{i+1} -> Exp(dt v_{i+1}) -> T_{i+2} -> Log_{i+2} -- and the
// step of knot i is
//     b_log (+ a_rho): tau_i = Log(T_nom^-1 T_i);   a_pre, a_post: u_i, v_{i+1} = v_i + dt a(q_i, v_i, u_i);
//     a_exp: E_{i+1} = Exp(dt v_{i+1});             b_compose: T_{i+2} = T_{i+1} E_{i+1}
// (the a_ / b_ prefixes are those of an earlier partition of the knot into a control and a pose wavefront).

// Exp(dt v) (quadrotor_model.cc:33-49, 174-200 -- the pose integrates with the OLD velocity): the quaternion
// increment DQ = (sin(th/2)/th theta_e ; cos(th/2)) as (x, y, z, w) in every quad, the translation increment PP in Q1
template <class W>
QILQR_HD void a_exp(const RConsts<W> &k, typename W::V VL, typename W::V VW, typename W::V &DQ, typename W::V &PP) {
  typedef typename W::V V;
  typedef typename W::M M;
  const double eps = Eps<double>::manif;
  // ---- Exp coefficients: cos(th/2), sin(th/2)/th, (1 - cos th)/th^2, (th - sin th)/th^3 in lanes 0..3 of a row
  const V THE = k.dt * VW;
  const V sqe = THE * THE;
  V th2e = W::template bc<0>(sqe) + W::template bc<1>(sqe);
  th2e = th2e + W::template bc<2>(sqe);
  V P1;
  {
    const V x = th2e, x2 = x * x, x4 = x2 * x2;
    const V p01 = W::fma(k.PC[1], x, k.PC[0]), p23 = W::fma(k.PC[3], x, k.PC[2]), p45 = W::fma(k.PC[5], x, k.PC[4]),
            p67 = W::fma(k.PC[7], x, k.PC[6]);
    P1 = W::fma(W::fma(p67, x2, p45), x4, W::fma(p23, x2, p01));
    // manif's branches of Exp: small angle, closed forms beyond the series' range; patch lanes 0..3 of P1.  The closed
    // forms (sqrt, two sin, two cos) run only when some row is beyond the series' range, the small-angle constants only
    // when some row is at rest: a converging solve takes neither branch.
    const M small = W::lnot(W::gt(th2e, eps));
    const M closed = W::gt(th2e, Series<double>::EXP_MAX);
    if (QILQR_RARE(W::any(closed))) P1 = W::exp_closed(closed, th2e, P1, k.L0, k.L1, k.L2, k.L3);
    if (QILQR_RARE(W::any(small))) {
      P1 = W::sel(W::land(small, k.L0), V(1.0), P1);
      P1 = W::sel(W::land(small, W::lor(k.L1, k.L2)), V(0.5), P1);
      P1 = W::sel(W::land(small, k.L3), V(0.0), P1);
    }
  }
  // ---- p = rho_e + a theta_e x rho_e + b theta_e x (theta_e x rho_e) in Q1
  const V A1 = k.SA * VW;    // [0 | theta_e | . | 0]
  const V B1 = k.SB1 * VL;   // [0 | rho_e   | 0 | 0]
  const V A1p = W::rot1(A1), A1pp = W::rot1(A1p);
  const V C1 = cross_rot<W>(A1p, A1pp, B1);
  const V C2 = cross_rot<W>(A1p, A1pp, C1);
  const V CA = W::template bc<2>(P1) * k.MQ1, CB = W::template bc<3>(P1) * k.MQ1;
  PP = W::fma(CB, C2, W::fma(CA, C1, B1));
  const V CH = W::template bc<0>(P1), SH = W::template bc<1>(P1);
  DQ = W::fma(k.M3, SH * THE, k.MW * CH);
}
// the closed forms of a_exp (out of line on the device: rarely taken, and large)
template <class W>
QILQR_HD typename W::V exp_closed_forms(typename W::M closed, typename W::V th2e, typename W::V P1, typename W::M L0, typename W::M L1,
                                        typename W::M L2, typename W::M L3) {
  typedef typename W::V V;
  typedef typename W::M M;
  // Round 6: between the eight-term series' range and EXP2_MAX (|dt omega| <= 3.46 rad per step) the same four coefficients as sixteen-term
  // series -- two Horner chains in x^2, a lane's coefficients by its place in the row -- instead of sqrt, two sines, two cosines and three
  // divisions.  The optimised trajectory of the reference's demo (configs[0]) spins through 45 of its 100 knots at |dt omega|^2 up to 6.1
  // in every rollout: 149.5 -> ~131 us per round of a lone trajectory.  Beyond EXP2_MAX: manif's closed forms, as before.
  {
    const M far = W::land(closed, W::gt(th2e, Series<double>::EXP2_MAX));
    const M mid = W::land(closed, W::lnot(far));
    if (W::any(mid)) {
      const V x = W::sel(mid, th2e, V(1.0)), x2 = x * x;
      // (W::exp2_coeff(k): coefficient k of the series of the lane's place in its row -- on the device a table in LDS, filled when the
      // kernel starts: per-lane loads from constant memory would cost the memory latency the closed forms cost in arithmetic)
      V ev = W::exp2_coeff(14), od = W::exp2_coeff(15);
#define QILQR_R16_EXP2_STEP(K)                   \
      ev = W::fma(ev, x2, W::exp2_coeff(K));     \
      od = W::fma(od, x2, W::exp2_coeff(K + 1));
      QILQR_R16_EXP2_STEP(12) QILQR_R16_EXP2_STEP(10) QILQR_R16_EXP2_STEP(8) QILQR_R16_EXP2_STEP(6) QILQR_R16_EXP2_STEP(4)
      QILQR_R16_EXP2_STEP(2) QILQR_R16_EXP2_STEP(0)
#undef QILQR_R16_EXP2_STEP
      const V series = W::fma(od, x, ev);
      P1 = W::sel(W::land(mid, W::lor(W::lor(L0, L1), W::lor(L2, L3))), series, P1);
    }
    if (!W::any(far)) return P1;
    closed = far;
  }
  const V xc = W::sel(closed, th2e, V(1.0));
  const V theta = W::sqrt_(xc), ha = 0.5 * theta;
  const V sn = W::sin_(ha), cs = W::cos_(ha), st = W::sin_(theta), ct = W::cos_(theta);
  P1 = W::sel(W::land(closed, L0), cs, P1);
  P1 = W::sel(W::land(closed, L1), sn / theta, P1);
  P1 = W::sel(W::land(closed, L2), (1.0 - ct) / xc, P1);
  return W::sel(W::land(closed, L3), (theta - st) / (xc * theta), P1);
}

// T <- T E with E = (DQ, PP) from a_exp:  t += R(q) p,  q <- q DQ
template <class W>
QILQR_HD void b_compose(const RConsts<W> &k, typename W::V TT, typename W::V QQ, typename W::V DQ, typename W::V PP,
                        typename W::V &TTn, typename W::V &QQn) {
  typedef typename W::V V;
  typedef typename W::M M;
  const double eps = Eps<double>::manif;
  const V QQp = W::rot1(QQ), QQpp = W::rot1(QQp);
  const V C3 = cross_rot<W>(QQp, QQpp, PP);   // Q1: u x p
  const V C4 = cross_rot<W>(QQp, QQpp, C3);   // Q1: u x (u x p)
  const V Wq = W::template bc<3>(QQ);
  const V Rp = W::fma(V(2.0), C4, W::fma(Wq + Wq, C3, PP));
  TTn = W::fma(k.MQ1_3, Rp, TT);
  // q d = (w d_v + d_w u + u x d_v ; w d_w - u . d_v)
  const V ZX = cross_rot<W>(QQp, QQpp, DQ);  // u x d_v in every quad (0 in lane j = 3)
  const V DW = W::template bc<3>(DQ);
  const V ov = W::fma(Wq, DQ, W::fma(DW, k.M3 * QQ, ZX));
  const V prod = QQ * DQ;
  V dq = W::template bc<0>(prod) + W::template bc<1>(prod);
  dq = dq + W::template bc<2>(prod);
  V Qn = W::fma(-k.MW, dq, ov);
  {
    const V sqq = Qn * Qn;
    V nq = (W::template bc<0>(sqq) + W::template bc<1>(sqq)) + (W::template bc<2>(sqq) + W::template bc<3>(sqq));
    const M off = W::gt(W::abs_(nq - 1.0), eps);  // manif's compose renormalisation
    if (QILQR_RARE(W::any(off))) Qn = Qn * W::sel(off, 2.0 / (1.0 + nq), V(1.0));
  }
  QQn = Qn;
}

// the pose wave, second half: the pose error tau = Log(T_nom^-1 T) = [rho ; theta] (quadrotor_model.cc:215-219;
// se3_rminus_fast) up to the last two cross products, which the control wave does (a_rho): TH4 = theta in lanes j < 3 of Q0 and
// the Jacobian coefficient c in lane j = 3, TD = R_n^T (t - t_n) in Q0
template <class W>
QILQR_HD void b_log(const RConsts<W> &k, typename W::V TT, typename W::V QQ, const typename W::V *op, typename W::V &TH4,
                    typename W::V &TD) {
  typedef typename W::V V;
  typedef typename W::M M;
  const double eps = Eps<double>::manif;
  // ---- rotation part: qd = conj(q_n) q in Q0 (x, y, z, w)
  V qd = W::template dot4<0>(V(0.0), QQ, op[OP_LQ + 0], op[OP_LQ + 1], op[OP_LQ + 2], op[OP_LQ + 3]);
  V sq = qd * qd;
  V s2 = W::template bc<0>(sq) + W::template bc<1>(sq);
  s2 = s2 + W::template bc<2>(sq);
  {
    const V nn = s2 + W::template bc<3>(sq);
    const M off = W::gt(W::abs_(nn - 1.0), eps);  // manif's compose renormalisation
    if (QILQR_RARE(W::any(off))) {
      const V sc = W::sel(off, 2.0 / (1.0 + nn), V(1.0));
      qd = qd * sc;
      sq = qd * qd;
      s2 = W::template bc<0>(sq) + W::template bc<1>(sq);
      s2 = s2 + W::template bc<2>(sq);
    }
  }
  const V wq = W::template bc<3>(qd);
  // ---- 2 asin(s)/s: its two halves side by side in lanes 4, 5 of a row
  V coeff;
  {
    const V x = s2, x2 = x * x, x4 = x2 * x2;
    const V p01 = W::fma(k.PC[1], x, k.PC[0]), p23 = W::fma(k.PC[3], x, k.PC[2]), p45 = W::fma(k.PC[5], x, k.PC[4]),
            p67 = W::fma(k.PC[7], x, k.PC[6]);
    const V P1 = W::fma(W::fma(p67, x2, p45), x4, W::fma(p23, x2, p01));
    const V y8 = x4 * x4;
    coeff = W::template fm<5>(W::template bc<4>(P1), P1, y8);
    // manif's branches of Log (so3_log): small angle, and the atan2 form outside the series' range / for w <= 0
    const M small = W::lnot(W::gt(s2, eps));
    const M series = W::land(W::gt(wq, 0.0), W::lnot(W::gt(s2, Series<double>::LOG_MAX)));
    const M closed = W::land(W::lnot(small), W::lnot(series));
    if (QILQR_RARE(W::any(closed))) coeff = W::log_closed(closed, s2, wq, coeff);
    coeff = W::sel(small, V(2.0), coeff);
  }
  const V th = coeff * qd;  // Q0: theta (lane j = 3 holds coeff w: finite, never used)
  const V th2 = (coeff * coeff) * s2;
  V cJ;  // 1/th^2 - (1 + cos th)/(2 th sin th)
  {
    const double *jc = Series<double>::jinv_c;
    const V x2 = th2 * th2, x4 = x2 * x2;
    const V p01 = W::fma(V(jc[1]), th2, V(jc[0])), p23 = W::fma(V(jc[3]), th2, V(jc[2])), p45 = W::fma(V(jc[5]), th2, V(jc[4])),
            p67 = W::fma(V(jc[7]), th2, V(jc[6]));
    cJ = W::fma(W::fma(p67, x2, p45), x4, W::fma(p23, x2, p01));
    const M small = W::lnot(W::gt(th2, eps));  // a rollout that has converged onto its nominal trajectory is here at every knot
    const M closed = W::gt(th2, Series<double>::JINV_MAX);
    if (QILQR_RARE(W::any(closed))) cJ = W::jinv_closed(closed, th2, cJ);
    cJ = W::sel(small, V(0.0), cJ);
  }
  // ---- td = R_n^T (t - t_n) in Q0
  const V dT = TT - op[OP_TN];
  TD = W::template dot3<4>(V(0.0), dT, op[OP_RT + 0], op[OP_RT + 1], op[OP_RT + 2]);
  TH4 = W::fma(k.M3, th, k.MW * cJ);
}
// control wave: rho = td - theta x td / 2 + c theta x (theta x td) in Q0, from the pose wave's (TH4, TD)
template <class W>
QILQR_HD typename W::V a_rho(typename W::V TH4, typename W::V TD) {
  typedef typename W::V V;
  const V cJ = W::template bc<3>(TH4);
  const V thp = W::rot1(TH4), thpp = W::rot1(thp);  // (rot1 leaves lane j = 3 in place: it never reaches lanes j < 3)
  const V C1 = cross_rot<W>(thp, thpp, TD);
  const V C2 = cross_rot<W>(thp, thpp, C1);
  return W::fma(cJ, C2, W::fma(V(-0.5), C1, TD));
}

// the closed forms of b_log (out of line on the device)
template <class W>
QILQR_HD typename W::V log_closed_forms(typename W::M closed, typename W::V s2, typename W::V wq, typename W::V coeff) {
  typedef typename W::V V;
  typedef typename W::M M;
  const V ss = W::sqrt_(W::sel(closed, s2, V(1.0)));
  const M neg = W::lt(wq, 0.0);
  const V at = W::atan2_(W::sel(neg, -ss, ss), W::sel(neg, -wq, wq));
  return W::sel(closed, 2.0 * at / ss, coeff);
}
template <class W>
QILQR_HD typename W::V jinv_closed_forms(typename W::M closed, typename W::V th2, typename W::V cJ) {
  typedef typename W::V V;
  const V x = W::sel(closed, th2, V(1.0));
  const V theta = W::sqrt_(x);
  return W::sel(closed, 1.0 / x - (1.0 + W::cos_(theta)) / (2.0 * theta * W::sin_(theta)), cJ);
}

// control wave, the part of knot i that needs only v_i and the operands (it runs while the pose wave is still in Log_i):
// the velocity terms of the control law, the gyroscopic term, the velocity lanes of the stored knot
template <class W>
struct APre {
  typename W::V u2, u3, aW, st0;
};
template <class W>
QILQR_HD void a_pre(const RConsts<W> &k, typename W::V VL, typename W::V VW, const typename W::V *op, APre<W> &r) {
  typedef typename W::V V;
  const V dvl = VL - op[OP_VNL], dvw = VW - op[OP_VNW];
  r.u2 = W::template dot3<0>(V(0.0), dvl, op[OP_K + 6], op[OP_K + 7], op[OP_K + 8]);
  r.u3 = W::template dot3<8>(V(0.0), dvw, op[OP_K + 9], op[OP_K + 10], op[OP_K + 11]);  // (v_nom's angular part sits in Q2)
  r.st0 = W::fma(k.MQ2_3, VW, k.MQ0_3 * VL);
  const V IW = W::template dot3<0>(V(0.0), VW, k.IC[0], k.IC[1], k.IC[2]);  // I omega in Q2
  const V A1 = k.SA * VW;  // [0 | . | omega | 0]
  const V A1p = W::rot1(A1), A1pp = W::rot1(A1p);
  const V C1 = cross_rot<W>(A1p, A1pp, IW);  // Q2: omega x I omega
  r.aW = W::template dot3<8>(VW, C1, k.NI[0], k.NI[1], k.NI[2]);  // omega - dt I^-1 (omega x I omega)
}
// control wave, with (theta | c, td) of tau_i and q_i from the pose wave: u_i (ilqr.hh:158-161; rows in Q3), v_{i+1} = v_i + dt a(q_i,
// v_i, u_i) (quadrotor_model.cc:65-78), and what the wave stores for knot i: st = [v_lin | 0 | omega | u]
// (RHO_DONE: TD is rho already -- a_rho was applied by the wave that computed the Log)
template <class W, bool RHO_DONE = false>
QILQR_HD typename W::V a_post(const RConsts<W> &k, const APre<W> &r, typename W::V TH, typename W::V TD, typename W::V QQ,
                              typename W::V VL, const typename W::V *op, bool advance, typename W::V &VLn, typename W::V &VWn) {
  typedef typename W::V V;
  const V RH = RHO_DONE ? TD : a_rho<W>(TH, TD);  // (lane j = 3 of TH holds the Jacobian coefficient: the control law broadcasts lanes 0..2 only)
  const V u1 = W::template dot3<0>(V(0.0), TH, op[OP_K + 3], op[OP_K + 4], op[OP_K + 5]);
  const V u0 = W::template dot3<0>(op[OP_U0], RH, op[OP_K + 0], op[OP_K + 1], op[OP_K + 2]);
  const V UU = (u0 + u1) + (r.u2 + r.u3);  // valid in Q3 (the shared registers leave other operands' products elsewhere)
  const V st = W::fma(k.MQ3, UU, r.st0);
  if (!advance) return st;  // the reference's step after the last knot is computed and discarded (ilqr.hh:168)
  // gravity in the body frame: (z, w, x, .) and (w, z, y, .) of q
  const V aL = W::template dot2<0>(VL + k.GRAVC, QQ, k.S1 * W::template qperm<0xCE>(QQ), k.S2 * W::template qperm<0xDB>(QQ));
  // J_u u: rows 8..11 of the constant control Jacobian (thrust along body z, torques)
  VLn = W::template dot4<12>(aL, UU, k.GLz, k.GLz, k.GLz, k.GLz);
  VWn = W::template dot4<12>(r.aW, UU, k.GW[0], k.GW[1], k.GW[2], k.GW[3]);
  return st;
}

// Which element of the 18-double knot a lane of a stored register holds (-1: none).  The control wave stores
// st = [v_lin | - | omega | u]; the pose wave stores the translation TT (Q1) and the quaternion QQ (its Q0 copy:
// x, y, z, w -> elements 5, 6, 7, 4).
QILQR_HD int sta_elem(int l) {
  const int q = (l >> 2) & 3, j = l & 3;
  if (q == 3) return 14 + j;
  if (j == 3 || q == 1) return -1;
  return q == 0 ? 8 + j : 11 + j;
}
// pose = [q (x, y, z, w) in Q0 | t in Q1] (RConsts::MQ0 q + t: the translation register is zero outside Q1) in ONE store
QILQR_HD int stp_elem(int l) {
  const int q = (l >> 2) & 3, j = l & 3;
  if (q == 0) return j == 3 ? 4 : 5 + j;
  return (q == 1 && j < 3) ? 1 + j : -1;
}
QILQR_HD int stt_elem(int l) { return (((l >> 2) & 3) == 1 && (l & 3) < 3) ? 1 + (l & 3) : -1; }
QILQR_HD int stq_elem(int l) {
  const int q = (l >> 2) & 3, j = l & 3;
  if (q != 0) return -1;
  return j == 3 ? 4 : 5 + j;
}
// the state of knot 0: which element a lane loads into TT / QQ / VL / VW (-1: zero)
QILQR_HD int tt_elem(int l) { return (((l >> 2) & 3) == 1 && (l & 3) < 3) ? 1 + (l & 3) : -1; }
QILQR_HD int qq_elem(int l) { return (l & 3) == 3 ? 4 : 5 + (l & 3); }
QILQR_HD int vl_elem(int l) { return (l & 3) < 3 ? 8 + (l & 3) : -1; }
QILQR_HD int vw_elem(int l) { return (l & 3) < 3 ? 11 + (l & 3) : -1; }

// ---------------------------------------------------------------------------------------------------------------
// Operand preparation (wavefront P).  `ld(e)` returns element e of the nominal knot (18 doubles: time, t, q(w,x,y,z),
// v, u) of this lane's trajectory as a value per lane -- e may differ from lane to lane; `lg(e)` the same for the 52
// gains [k(4) | K column-major].  `alpha` is the trajectory's step size (value per lane, uniform in a row).
template <class W>
struct PConsts {
  typedef typename W::V V;
  typename W::I e_uj, e_h[3], e_lq[4], e_tn, e_vl, e_vw, e_un, e_k, e_K[12];  // element indices per lane
  V s_h[3], s_lq[4], d_c[3], mQ0_3, mQ0, mQ1_3, mQ2_3, mQ3, m3;               // signs / masks per lane
};
template <class W>
QILQR_HD void make_pconsts(PConsts<W> &p) {
  auto q = [](int l) { return (l >> 2) & 3; };
  auto j = [](int l) { return l & 3; };
  // elements of the knot: 1..3 t, 4 qw, 5..7 q(x, y, z), 8..10 v_lin, 11..13 omega, 14..17 u
  p.e_uj = W::iconst([&](int l) { return 5 + (j(l) < 3 ? j(l) : 0); });
  // row c of hat(u): H[c][j] = sign * u_idx:   row 0 = (0, -z, y), row 1 = (z, 0, -x), row 2 = (-y, x, 0)
  // (the index is the third of {0, 1, 2} beside c and j, the sign that of the permutation (c, j, index) -- written as arithmetic: a table
  // indexed by the lane put nine words into scratch memory, and P's first knot waited for them at the start of every rollout)
  auto hidx = [](int c, int jj) { return c == jj ? 0 : 3 - c - jj; };
  auto hsgn = [](int c, int jj) { return c == jj ? 0.0 : ((jj - c + 3) % 3 == 1 ? -1.0 : 1.0); };
  for (int c = 0; c < 3; ++c) {
    p.e_h[c] = W::iconst([&](int l) { return 5 + hidx(c, j(l) < 3 ? j(l) : 0); });
    p.s_h[c] = W::vconst([&](int l) { return (q(l) == 0 && j(l) < 3) ? hsgn(c, j(l)) : 0.0; });
    p.d_c[c] = W::vconst([&](int l) { return (q(l) == 0 && j(l) == c) ? 1.0 : 0.0; });
  }
  // left multiplication by a = conj(q_n) = (-x, -y, -z, w): rows (o_x, o_y, o_z, o_w), columns (b_x, b_y, b_z, b_w)
  //   o_x: ( aw, -az,  ay, ax)   o_y: ( az, aw, -ax, ay)   o_z: (-ay, ax, aw, az)   o_w: (-ax, -ay, -az, aw)
  // in terms of q_n (x, y, z, w) = elements (5, 6, 7, 4): a_v = -q_v
  const int lidx[4][4] = {{4, 7, 6, 5}, {7, 4, 5, 6}, {6, 5, 4, 7}, {5, 6, 7, 4}};      // [row][col] -> element
  const double lsgn[4][4] = {{1, 1, -1, -1}, {-1, 1, 1, -1}, {1, -1, 1, -1}, {1, 1, 1, 1}};
  for (int c = 0; c < 4; ++c) {
    p.e_lq[c] = W::iconst([&](int l) { return lidx[j(l)][c]; });
    p.s_lq[c] = W::vconst([&](int l) { return q(l) == 0 ? lsgn[j(l)][c] : 0.0; });
  }
  p.e_tn = W::iconst([&](int l) { return 1 + (j(l) < 3 ? j(l) : 0); });
  p.e_vl = W::iconst([&](int l) { return 8 + (j(l) < 3 ? j(l) : 0); });
  p.e_vw = W::iconst([&](int l) { return 11 + (j(l) < 3 ? j(l) : 0); });
  p.e_un = W::iconst([&](int l) { return 14 + j(l); });
  p.e_k = W::iconst([&](int l) { return j(l); });
  for (int c = 0; c < 12; ++c) p.e_K[c] = W::iconst([&](int l) { return 4 + 4 * c + j(l); });
  p.mQ0_3 = W::vconst([&](int l) { return (q(l) == 0 && j(l) < 3) ? 1.0 : 0.0; });
  p.mQ0 = W::vconst([&](int l) { return q(l) == 0 ? 1.0 : 0.0; });
  p.mQ1_3 = W::vconst([&](int l) { return (q(l) == 1 && j(l) < 3) ? 1.0 : 0.0; });
  p.mQ3 = W::vconst([&](int l) { return q(l) == 3 ? 1.0 : 0.0; });
  p.mQ2_3 = W::vconst([&](int l) { return (q(l) == 2 && j(l) < 3) ? 1.0 : 0.0; });
  p.m3 = W::vconst([&](int l) { return j(l) < 3 ? 1.0 : 0.0; });
}

// raw values a lane of P loads for one knot (its own element of each group), in the storage type of the loader
enum { RAW_X = 0, RAW_Y, RAW_Z, RAW_W, RAW_UJ, RAW_H = 5, RAW_TN = 8, RAW_LQ = 9, RAW_K = 13, RAW_KFF = 25, RAW_UN, RAW_VL, RAW_VW, NRAW };
// phase 1: request the loads (nothing here uses a loaded value: on the device the requests of knot k + 1 are in flight
// while knot k is being converted and written)
template <class W, class RAWT, class LD, class LG>
QILQR_HD void p_load(const PConsts<W> &p, LD ld, LG lg, RAWT *raw) {
  raw[RAW_X] = ld(W::iuni(5)); raw[RAW_Y] = ld(W::iuni(6)); raw[RAW_Z] = ld(W::iuni(7)); raw[RAW_W] = ld(W::iuni(4));
  raw[RAW_UJ] = ld(p.e_uj);
  for (int c = 0; c < 3; ++c) raw[RAW_H + c] = ld(p.e_h[c]);
  raw[RAW_TN] = ld(p.e_tn);
  for (int c = 0; c < 4; ++c) raw[RAW_LQ + c] = ld(p.e_lq[c]);
  for (int c = 0; c < 12; ++c) raw[RAW_K + c] = lg(p.e_K[c]);
  raw[RAW_KFF] = lg(p.e_k);
  raw[RAW_UN] = ld(p.e_un);
  raw[RAW_VL] = ld(p.e_vl);
  raw[RAW_VW] = ld(p.e_vw);
}
// phase 2: the operand registers from the raw values
template <class W, class RAWT>
QILQR_HD void p_compute(const PConsts<W> &p, const RAWT *raw, typename W::V alpha, typename W::V *op) {
  typedef typename W::V V;
  // nominal rotation R_n = I + 2 w hat(u) + 2 hat(u)^2, row c as a lane vector over j:
  //   R_n[c][j] = 2 u_c u_j + 2 w hat(u)[c][j] + delta_cj (1 - 2 |u|^2)
  const V x = V(raw[RAW_X]), y = V(raw[RAW_Y]), z = V(raw[RAW_Z]), w = V(raw[RAW_W]);
  const V uj = V(raw[RAW_UJ]) * p.mQ0_3;
  const V dd = 1.0 - 2.0 * ((x * x + y * y) + z * z);
  const V w2 = w + w;
  const V uc[3] = {x + x, y + y, z + z};
  V rt[3], lq[4];
  for (int c = 0; c < 3; ++c) {
    const V h = V(raw[RAW_H + c]) * p.s_h[c];
    rt[c] = W::fma(uc[c], uj, W::fma(w2, h, p.d_c[c] * dd));
  }
  for (int c = 0; c < 4; ++c) lq[c] = V(raw[RAW_LQ + c]) * p.s_lq[c];
  for (int c = 0; c < 12; ++c) {
    const V kq3 = V(raw[RAW_K + c]) * p.mQ3;
    op[OP_K + c] = c < 3 ? kq3 + rt[c] : (c < 7 ? kq3 + lq[c - 3] : kq3);  // (Q0 of the first seven: R_n^T, conj(q_n))
  }
  op[OP_MISC] = W::fma(alpha, V(raw[RAW_KFF]), V(raw[RAW_UN])) * p.mQ3 + V(raw[RAW_TN]) * p.mQ1_3 +
                (V(raw[RAW_VL]) * p.mQ0_3 + V(raw[RAW_VW]) * p.mQ2_3);
}

}  // namespace r16
}  // namespace qilqr
