// host_harness.cpp -- CPU test harness (tests only, never part of the product library).
//
// Compiles the lane-local device functions of quadrotorilqr_amd/csrc/se3_math.h with g++ and
// re-enacts k_backward's 64-lane data flow on the CPU (the MFMA and the cross-lane shuffles
// are emulated from their documented lane maps) so that `pytest -m "not gpu"` can check the
// knot records, the rollout and the operand layout of the backward pass against the oracle
// before anything runs on a GPU.
#include <cmath>
#include <cstring>
#include <vector>

#include "../quadrotorilqr_amd/csrc/backward_layout.h"
#include "../quadrotorilqr_amd/csrc/host_model.h"
#include "../quadrotorilqr_amd/csrc/rollout16.h"
#include "../quadrotorilqr_amd/csrc/se3_math.h"

using namespace qilqr;

namespace {
// v_mfma_f64_16x16x4_f64: lane l = (j = l & 15, kk = l >> 4) supplies A[j][kk], B[kk][j];
// result register r of lane l is D[4 r + kk][j]
void mfma_f64_16x16x4(const double a[64], const double b[64], double acc[64][4]) {
  double A[16][4], Bm[4][16], D[16][16];
  for (int l = 0; l < 64; ++l) {
    A[l & 15][l >> 4] = a[l];
    Bm[l >> 4][l & 15] = b[l];
  }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = acc[j + 16 * (i & 3)][i >> 2];
      for (int k = 0; k < 4; ++k) s += A[i][k] * Bm[k][j];
      D[i][j] = s;
    }
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) acc[l][r] = D[4 * r + (l >> 4)][l & 15];
}

// ---------------------------------------------------------------------------------------------------------------
// A wavefront on the CPU for rollout16.h: 64 lanes per value, the cross-lane operations from their definitions
// (row_newbcast: lane L of the caller's row of 16; quad_perm: lane 4q + i reads lane 4q + ((ctrl >> 2 i) & 3)).
struct HV {
  double v[64];
  HV() { for (double &x : v) x = 0.0; }
  HV(double s) { for (double &x : v) x = s; }
};
struct HM { bool b[64]; };
struct HI { int v[64]; };
#define HV_BIN(op)                                                                                       \
  inline HV operator op(const HV &a, const HV &b) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = a.v[l] op b.v[l]; return r; } \
  inline HV operator op(const HV &a, double b) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = a.v[l] op b; return r; }         \
  inline HV operator op(double a, const HV &b) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = a op b.v[l]; return r; }
HV_BIN(+) HV_BIN(-) HV_BIN(*) HV_BIN(/)
inline HV operator-(const HV &a) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = -a.v[l]; return r; }
struct HostWave {
  typedef HV V;
  typedef HM M;
  typedef HI I;
  template <class F> static V vconst(F f) { V r; for (int l = 0; l < 64; ++l) r.v[l] = f(l); return r; }
  template <class F> static M mconst(F f) { M r; for (int l = 0; l < 64; ++l) r.b[l] = f(l); return r; }
  template <class F> static I iconst(F f) { I r; for (int l = 0; l < 64; ++l) r.v[l] = f(l); return r; }
  static I iuni(int e) { I r; for (int l = 0; l < 64; ++l) r.v[l] = e; return r; }
  template <int L> static V bc(const V &x) { V r; for (int l = 0; l < 64; ++l) r.v[l] = x.v[(l & ~15) + L]; return r; }
  template <int L> static V fm(const V &acc, const V &src, const V &m) { return fma(bc<L>(src), m, acc); }
  // acc + sum_c src[lane L0 + c of the row] * m_c, accumulated in this order (on the device: a chain of v_fmac_f64_dpp)
  template <int L0> static V dot2(const V &acc, const V &src, const V &m0, const V &m1) { return fm<L0 + 1>(fm<L0>(acc, src, m0), src, m1); }
  template <int L0> static V dot3(const V &acc, const V &src, const V &m0, const V &m1, const V &m2) {
    return fm<L0 + 2>(dot2<L0>(acc, src, m0, m1), src, m2);
  }
  template <int L0> static V dot4(const V &acc, const V &src, const V &m0, const V &m1, const V &m2, const V &m3) {
    return fm<L0 + 3>(dot3<L0>(acc, src, m0, m1, m2), src, m3);
  }
  template <int CTRL> static V qperm(const V &x) {
    V r;
    for (int l = 0; l < 64; ++l) r.v[l] = x.v[(l & ~3) + ((CTRL >> (2 * (l & 3))) & 3)];
    return r;
  }
  static V rot1(const V &x) { return qperm<0xC9>(x); }
  static V fma(const V &a, const V &b, const V &c) { V r; for (int l = 0; l < 64; ++l) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]); return r; }
  static bool any(const M &m) { for (bool x : m.b) if (x) return true; return false; }
  static V sel(const M &m, const V &a, const V &b) { V r; for (int l = 0; l < 64; ++l) r.v[l] = m.b[l] ? a.v[l] : b.v[l]; return r; }
  static M gt(const V &a, const V &b) { M r; for (int l = 0; l < 64; ++l) r.b[l] = a.v[l] > b.v[l]; return r; }
  static M lt(const V &a, const V &b) { M r; for (int l = 0; l < 64; ++l) r.b[l] = a.v[l] < b.v[l]; return r; }
  static M land(const M &a, const M &b) { M r; for (int l = 0; l < 64; ++l) r.b[l] = a.b[l] && b.b[l]; return r; }
  static M lor(const M &a, const M &b) { M r; for (int l = 0; l < 64; ++l) r.b[l] = a.b[l] || b.b[l]; return r; }
  static M lnot(const M &a) { M r; for (int l = 0; l < 64; ++l) r.b[l] = !a.b[l]; return r; }
#define HV_FUN(name, fn) static V name(const V &a) { V r; for (int l = 0; l < 64; ++l) r.v[l] = fn(a.v[l]); return r; }
  HV_FUN(abs_, std::fabs) HV_FUN(sqrt_, std::sqrt) HV_FUN(sin_, std::sin) HV_FUN(cos_, std::cos)
  static V atan2_(const V &a, const V &b) { V r; for (int l = 0; l < 64; ++l) r.v[l] = std::atan2(a.v[l], b.v[l]); return r; }
  // the rarely-taken closed forms (out of line on the device)
  static V exp2_coeff(int k) { return vconst([k](int l) { return qilqr::Series<double>::exp2[l & 3][k]; }); }
  static V exp_closed(const M &c, const V &x, const V &p, const M &l0, const M &l1, const M &l2, const M &l3) {
    return qilqr::r16::exp_closed_forms<HostWave>(c, x, p, l0, l1, l2, l3);
  }
  static V log_closed(const M &c, const V &s2, const V &wq, const V &coeff) { return qilqr::r16::log_closed_forms<HostWave>(c, s2, wq, coeff); }
  static V jinv_closed(const M &c, const V &th2, const V &cJ) { return qilqr::r16::jinv_closed_forms<HostWave>(c, th2, cJ); }
};
}  // namespace

extern "C" {

int hh_make_consts(double mass, const double *inertia, double arm, double ttr, double g, const double *Q,
                   const double *R, double dt, ModelConsts<double> *out) {
  return make_model_consts(mass, inertia, arm, ttr, g, Q, R, dt, out) ? 0 : 1;
}
int hh_consts_size() { return (int)sizeof(ModelConsts<double>); }
int hh_lin_stride() { return LIN_MAX_STRIDE; }
// layout chosen exactly as qilqr_create does; out = {sym, ur_zero, off_cxx, off_g, off_cost, stride}
void hh_layout(const ModelConsts<double> *c, int force_general, int *out) {
  bool qsym = true, ur0 = true;
  for (int i = 0; i < 12; ++i)
    for (int k = 0; k < 12; ++k) {
      qsym = qsym && (c->Q[i * 12 + k] == c->Q[k * 12 + i]);
      if (i < 6 && k >= 6) ur0 = ur0 && (c->Q[i * 12 + k] == 0.0);
    }
  const RecLayout L = make_layout(qsym && !force_general, ur0);
  out[0] = L.sym; out[1] = L.ur_zero; out[2] = L.off_cxx; out[3] = L.off_g; out[4] = L.off_cost; out[5] = L.stride;
}
static RecLayout layout_from(const int *v) {
  RecLayout L;
  L.sym = v[0]; L.ur_zero = v[1]; L.off_cxx = v[2]; L.off_g = v[3]; L.off_cost = v[4]; L.stride = v[5];
  L.tiled = 0;  // (the host harness keeps records contiguous: placement in memory is the device's business)
  L.dense_m = (v[2] == LIN_M_DENSE) ? 1 : 0;  // (the six-int wire form of the tests: the dense layouts are the ones whose C_xx starts behind a dense M)
  return L;
}
// the Runge-Kutta extension's layout for the same weights: out as hh_layout
void hh_layout_rk4(const ModelConsts<double> *c, int force_general, int *out) {
  int e[6];
  hh_layout(c, force_general, e);
  const RecLayout L = make_layout(e[0] != 0, e[1] != 0, true);
  out[0] = L.sym; out[1] = L.ur_zero; out[2] = L.off_cxx; out[3] = L.off_g; out[4] = L.off_cost; out[5] = L.stride;
}
// one Runge-Kutta step of the device code (se3_math.h, rk4_step): x = [t(3), q(w,x,y,z), v(6)] -> xn, M = [J_x | J_u] (12 x 16)
void hh_rk4_step(const ModelConsts<double> *c, const double *x, const double *u, double *xn, double *MU) {
  double t[3] = {x[0], x[1], x[2]}, q[4] = {x[4], x[5], x[6], x[3]}, v[6];
  for (int i = 0; i < 6; ++i) v[i] = x[7 + i];
  rk4_step(*c, t, q, v, u, MU);
  xn[0] = t[0]; xn[1] = t[1]; xn[2] = t[2];
  xn[3] = q[3]; xn[4] = q[0]; xn[5] = q[1]; xn[6] = q[2];
  for (int i = 0; i < 6; ++i) xn[7 + i] = v[i];
}
void hh_rollout_rk4(const ModelConsts<double> *c, const double *traj, const double *gains, double alpha, double *out, int n) {
  rollout_problem<false, double, 1>(*c, traj, gains, alpha, out, n);
}
// the table-indexed operand sources of k_backward against the value-returning ones; returns mismatches
int hh_check_operand_tables(const ModelConsts<double> *c, const int *lay) {
  const RecLayout L = layout_from(lay);
  double tab[CTAB_SIZE];
  build_ctab(c->Bu, c->Q, tab);
  int bad = 0;
  for (int r = 0; r < 12; ++r) {
    for (int col = 0; col < 16; ++col) {
      double cst;
      const int a = m_source(r, col, c->Bu, &cst), b = m_source_tab(r, col);
      if (a >= 0 ? (a != b) : (b >= 0 || tab[-1 - b] != cst)) ++bad;
    }
    for (int col = 0; col < 12; ++col) {
      double cst;
      const int a = cxx_source(L, r, col, c->Q, &cst), b = cxx_source_tab(L, r, col);
      if (a >= 0 ? (a != b) : (b >= 0 || tab[-1 - b] != cst)) ++bad;
    }
  }
  return bad;
}
// dense C_xx (12x12) rebuilt from a record through cxx_source()
void hh_dense_cxx(const ModelConsts<double> *c, const int *lay, const double *rec, double *Cxx) {
  const RecLayout L = layout_from(lay);
  for (int r = 0; r < 12; ++r)
    for (int col = 0; col < 12; ++col) {
      double cst;
      const int off = cxx_source(L, r, col, c->Q, &cst);
      Cxx[r * 12 + col] = off >= 0 ? rec[off] : cst;
    }
}

void hh_linearize(const ModelConsts<double> *c, const int *lay, const double *traj, const double *desired, int n,
                  double *lin) {
  const RecLayout L = layout_from(lay);
  for (int i = 0; i < n; ++i) linearize_knot(*c, L, traj + i * 18, desired + i * 18, lin + (long)i * L.stride);
}
// the tiled placement of the records (se3_math.h: rec_base / rec_elem, TiledRecWriter's paired stores): B trajectories of n knots
// are linearised through the tiled writer into `tiled` (rec_count(B, n, stride) doubles, pre-filled by the caller) and read
// back entry by entry into the plain [B][n][stride] array `lin`; returns TILE
int hh_linearize_tiled(const ModelConsts<double> *c, const int *lay, const double *traj, const double *desired, int B, int n,
                       double *tiled, double *lin) {
  RecLayout L = layout_from(lay);
  L.tiled = 1;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < n; ++i) {
      const double *pt = traj + ((long)b * n + i) * 18, *pd = desired + (long)i * 18;
      const TiledRecWriter<double> w{tiled + rec_base(L, b, n) + rec_elem(L, i, 0)};
      linearize_dynamics(*c, pt, w);
      w.flush();
      switch (layout_kind(L)) {
        case 0: linearize_cost<0>(c->Q, c->R, pt, pd, w); break;
        case 1: linearize_cost<1>(c->Q, c->R, pt, pd, w); break;
        default: linearize_cost<2>(c->Q, c->R, pt, pd, w); break;
      }
      w.flush();
    }
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < L.stride; ++k)
        lin[((long)b * n + i) * L.stride + k] = tiled[rec_base(L, b, n) + rec_elem(L, i, k)];
  return TILE;
}
long hh_rec_count(int B, int n, int stride) { return rec_count(B, n, stride); }
// the cost half of one knot through the instantiation `kind` of linearize_cost (0 general, 1 symmetric, 2 block-diagonal,
// 3 diagonal Q: the record of 2); rec: LIN_MAX_STRIDE doubles, pre-filled by the caller; returns the knot cost
double hh_linearize_cost_kind(const ModelConsts<double> *c, int kind, const double *pt, const double *pd, double *rec) {
  PlainRecWriter<double> w{rec};
  switch (kind) {
    case 0: return linearize_cost<0>(c->Q, c->R, pt, pd, w);
    case 1: return linearize_cost<1>(c->Q, c->R, pt, pd, w);
    case 2: return linearize_cost<2>(c->Q, c->R, pt, pd, w);
    default: return linearize_cost<3>(c->Q, c->R, pt, pd, w);
  }
}
void hh_rollout(const ModelConsts<double> *c, const double *traj, const double *gains, double alpha, double *out,
                int n) {
  rollout_problem<false>(*c, traj, gains, alpha, out, n);
}
// pose (-) pose and pose (+) tangent through the rollout's fast arithmetic; poses are [t ; q(w,x,y,z)]
void hh_rminus_fast(const double *Y, const double *X, double *tau) {
  const double qy[4] = {Y[4], Y[5], Y[6], Y[3]}, qx[4] = {X[4], X[5], X[6], X[3]};
  se3_rminus_fast(Y, qy, X, qx, tau);
}
void hh_rplus_fast(const double *Tin, const double *tau, double *Tout) {
  double t[3] = {Tin[0], Tin[1], Tin[2]}, q[4] = {Tin[4], Tin[5], Tin[6], Tin[3]};
  se3_rplus_fast(t, q, tau);
  Tout[0] = t[0]; Tout[1] = t[1]; Tout[2] = t[2];
  Tout[3] = q[3]; Tout[4] = q[0]; Tout[5] = q[1]; Tout[6] = q[2];
}
// the tiled layout's index arithmetic: scatter a plain [B][n][W] array into tiles, run the TILED
// rollout on trajectory b, and hand back its output in the plain layout
void hh_rollout_tiled(const ModelConsts<double> *c, const double *traj, const double *gains, const double *alpha,
                      double *out, int B, int n) {
  std::vector<double> tt(tiled_count(B, n, 18)), tg(tiled_count(B, n, 52)), to(tiled_count(B, n, 18));
  for (long b = 0; b < B; ++b)
    for (long i = 0; i < n; ++i) {
      for (int e = 0; e < 18; ++e) tt[knot_base<true>(b, n, 18) + knot_elem<true>(i, e, 18)] = traj[(b * n + i) * 18 + e];
      for (int e = 0; e < 52; ++e) tg[knot_base<true>(b, n, 52) + knot_elem<true>(i, e, 52)] = gains[(b * n + i) * 52 + e];
    }
  for (long b = 0; b < B; ++b)
    rollout_problem<true>(*c, tt.data() + knot_base<true>(b, n, 18), tg.data() + knot_base<true>(b, n, 52), alpha[b],
                          to.data() + knot_base<true>(b, n, 18), n);
  for (long b = 0; b < B; ++b)
    for (long i = 0; i < n; ++i)
      for (int e = 0; e < 18; ++e) out[(b * n + i) * 18 + e] = to[knot_base<true>(b, n, 18) + knot_elem<true>(i, e, 18)];
}

// k_rollout16 on the CPU: one wavefront = four trajectories (plain [4][n][18] / [4][n][52] arrays), the operand
// registers of every knot prepared by r16::p_load / p_compute and consumed by r16::b_log / a_pre / a_post / a_exp / b_compose
// exactly as the three device wavefronts do (the LDS ring and its flags are plumbing, not arithmetic: not re-enacted).  ops_out (optional):
// the 23 x 64 operand values of knot `ops_knot`.
void hh_rollout16(const ModelConsts<double> *c, const double *traj, const double *gains, const double *alpha, double *out,
                  int n, int ops_knot, double *ops_out) {
  using namespace r16;
  RConsts<HostWave> kc;
  make_rconsts(*c, kc);
  PConsts<HostWave> pc;
  make_pconsts(pc);
  auto T = [&](int l, int i, int e) { return traj[((long)(l >> 4) * n + i) * 18 + e]; };
  auto G = [&](int l, int i, int e) { return gains[((long)(l >> 4) * n + i) * 52 + e]; };
  HV TT, QQ, VL, VW;
  for (int l = 0; l < 64; ++l) {
    TT.v[l] = tt_elem(l) >= 0 ? T(l, 0, tt_elem(l)) : 0.0;
    QQ.v[l] = T(l, 0, qq_elem(l));
    VL.v[l] = vl_elem(l) >= 0 ? T(l, 0, vl_elem(l)) : 0.0;
    VW.v[l] = vw_elem(l) >= 0 ? T(l, 0, vw_elem(l)) : 0.0;
  }
  HV al;
  for (int l = 0; l < 64; ++l) al.v[l] = alpha[l >> 4];
  for (int i = 0; i < n; ++i) {
    auto ld = [&](const HI &e) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = T(l, i, e.v[l]); return r; };
    auto lg = [&](const HI &e) { HV r; for (int l = 0; l < 64; ++l) r.v[l] = G(l, i, e.v[l]); return r; };
    HV op[NOPS], raw[NRAW];
    p_load<HostWave>(pc, ld, lg, raw);
    p_compute<HostWave>(pc, raw, al, op);
    if (ops_out && i == ops_knot)
      for (int r = 0; r < NOPS; ++r)
        for (int l = 0; l < 64; ++l) ops_out[r * 64 + l] = op[r].v[l];
    // wave B has produced (t, q)_i and its Log against the nominal knot i; wave A takes it from there
    HV RH, TH, VLn, VWn;
    APre<HostWave> pre;
    a_pre<HostWave>(kc, VL, VW, op, pre);
    b_log<HostWave>(kc, TT, QQ, op, RH, TH);
    const HV st = a_post<HostWave>(kc, pre, RH, TH, QQ, VL, op, i + 1 < n, VLn, VWn);  // (RH, TH) here: b_log's (TH4, TD)
    for (int l = 0; l < 64; ++l) {
      double *o = out + ((long)(l >> 4) * n + i) * 18;
      if ((l & 15) == 0) o[0] = T(l, i, 0);
      if (sta_elem(l) >= 0) o[sta_elem(l)] = st.v[l];
      if (stt_elem(l) >= 0) o[stt_elem(l)] = TT.v[l];
      if (stq_elem(l) >= 0) o[stq_elem(l)] = QQ.v[l];
    }
    if (i + 1 < n) {  // the reference's step after the last knot is computed and discarded (ilqr.hh:168)
      HV TTn, QQn, DQ, PP;
      a_exp<HostWave>(kc, VL, VW, DQ, PP);  // E_i from v_i (on the device: computed by A one knot earlier)
      b_compose<HostWave>(kc, TT, QQ, DQ, PP, TTn, QQn);
      TT = TTn; QQ = QQn; VL = VLn; VW = VWn;
    }
  }
}
// dense J_x (12x12) and J_u (12x4) rebuilt from a knot record through m_source()
void hh_dense_jacobians(const ModelConsts<double> *c, const double *rec, double *Jx, double *Ju) {
  for (int r = 0; r < 12; ++r)
    for (int col = 0; col < 16; ++col) {
      double cst;
      const int off = m_source(r, col, c->Bu, &cst);
      const double v = off >= 0 ? rec[off] : cst;
      if (col < 12) Jx[r * 12 + col] = v;
      else Ju[r * 4 + (col - 12)] = v;
    }
}

// the same for a record of layout `lay` (the Runge-Kutta extension's dense M included)
void hh_dense_jacobians_lay(const ModelConsts<double> *c, const int *lay, const double *rec, double *Jx, double *Ju) {
  const RecLayout L = layout_from(lay);
  for (int r = 0; r < 12; ++r)
    for (int col = 0; col < 16; ++col) {
      double cst;
      const int off = m_source(L, r, col, c->Bu, &cst);
      const double v = off >= 0 ? rec[off] : cst;
      if (col < 12) Jx[r * 12 + col] = v;
      else Ju[r * 4 + (col - 12)] = v;
      if ((off >= 0 ? off : -1) != (m_source_tab(L, r, col) >= 0 ? m_source_tab(L, r, col) : -1)) Jx[0] = 0.0 / 0.0;  // the two forms must agree
    }
}

// the general kernel's factorisation of Q_uu on its own: x = A^-1 b by the device code's restatement of Eigen's pivoted LDL^T
void hh_ldlt4_pivoted_solve(const double *A, const double *b, double *x) {
  double Am[16], bv[4], xv[4];
  for (int e = 0; e < 16; ++e) Am[e] = A[e];
  for (int e = 0; e < 4; ++e) bv[e] = b[e];
  ldlt4_pivoted_solve(Am, bv, xv);
  for (int e = 0; e < 4; ++e) x[e] = xv[e];
}

// k_backward re-enacted lane by lane (same statements, loops over the 64 lanes between the
// points where the kernel exchanges data).  sym = 1: the SYM = true instantiation (accumulator tile
// reused as A operand, V_x / V_xx / k^T Quu k in their Q_xu forms), sym = 0: the general one.
void hh_backward_emulated(const ModelConsts<double> *cp, const int *lay, const double *lin, int n, double *gains,
                          double *terms, int sym) {
  const ModelConsts<double> &c = *cp;
  const RecLayout L = layout_from(lay);
  constexpr int LD = 17;
  double Vs[12 * LD] = {0}, Hs[16 * LD] = {0}, gs[16] = {0}, vxs[12] = {0};
  int moff[64][3], coff[64][3];
  double mconst[64][3], cconst[64][3], cuu[64];
  double va[64][3] = {{0}}, vxl[64][3] = {{0}};
  double QuTk[64] = {0}, kTQuuk[64] = {0};
  for (int l = 0; l < 64; ++l) {
    const int j = l & 15, kk = l >> 4;
    for (int kc = 0; kc < 3; ++kc) moff[l][kc] = m_source(4 * kc + kk, j, c.Bu, &mconst[l][kc]);
    for (int r = 0; r < 3; ++r) {
      cconst[l][r] = 0.0;
      coff[l][r] = (j < 12) ? cxx_source(L, 4 * r + kk, j, c.Q, &cconst[l][r]) : -1;
    }
    cuu[l] = (j >= 12) ? 2.0 * c.R[kk * 4 + (j - 12)] : 0.0;
  }
  for (int i = n - 1; i >= 0; --i) {
    const double *rec = lin + (long)i * L.stride;
    double m[64][3], cx[64][3], gcj[64], T[64][4], H[64][4], part[64], ghat[64];
    for (int l = 0; l < 64; ++l) {
      for (int kc = 0; kc < 3; ++kc) m[l][kc] = moff[l][kc] >= 0 ? rec[moff[l][kc]] : mconst[l][kc];
      for (int r = 0; r < 3; ++r) cx[l][r] = coff[l][r] >= 0 ? rec[coff[l][r]] : cconst[l][r];
      gcj[l] = rec[L.off_g + (l & 15)];
      for (int r = 0; r < 4; ++r) T[l][r] = 0.0;
    }
    double a[64], b[64];
    for (int kc = 0; kc < 3; ++kc) {
      for (int l = 0; l < 64; ++l) { a[l] = va[l][kc]; b[l] = m[l][kc]; }
      mfma_f64_16x16x4(a, b, T);
    }
    for (int l = 0; l < 64; ++l) { H[l][0] = cx[l][0]; H[l][1] = cx[l][1]; H[l][2] = cx[l][2]; H[l][3] = cuu[l]; }
    for (int kc = 0; kc < 3; ++kc) {
      for (int l = 0; l < 64; ++l) { a[l] = m[l][kc]; b[l] = T[l][kc]; }
      mfma_f64_16x16x4(a, b, H);
    }
    for (int l = 0; l < 64; ++l) part[l] = m[l][0] * vxl[l][0] + m[l][1] * vxl[l][1] + m[l][2] * vxl[l][2];
    {
      double t[64];
      for (int l = 0; l < 64; ++l) t[l] = part[l] + part[l ^ 16];
      for (int l = 0; l < 64; ++l) part[l] = t[l] + t[l ^ 32];
    }
    for (int l = 0; l < 64; ++l) {
      const int j = l & 15, kk = l >> 4;
      ghat[l] = gcj[l] + part[l];
      for (int r = 0; r < 4; ++r) Hs[(4 * r + kk) * LD + j] = H[l][r];
      if (kk == 0) gs[j] = ghat[l];
    }
    double aop[64], bop[64], vx[64], kcol_all[64][4], mc_all[64][4], Quu_l[16] = {0};
    for (int l = 0; l < 64; ++l) {
      const int j = l & 15, kk = l >> 4;
      double Quu[16], Qu[4], rhs[4];
      for (int aa = 0; aa < 4; ++aa) {
        for (int bb = 0; bb < 4; ++bb) Quu[aa * 4 + bb] = Hs[(12 + aa) * LD + 12 + bb];
        Qu[aa] = gs[12 + aa];
        if (sym) rhs[aa] = (j == 12) ? Qu[aa] : Hs[(12 + aa) * LD + j];  // col[aa] = H[12 + aa][j]
        else rhs[aa] = (j < 12) ? Hs[j * LD + 12 + aa] : ((j == 12) ? Qu[aa] : 0.0);  // lane 12 solves for k
      }
      const double i0 = 1.0 / Quu[0];
      const double l10 = Quu[4] * i0, l20 = Quu[8] * i0, l30 = Quu[12] * i0;
      const double d1 = Quu[5] - l10 * Quu[4];
      const double i1 = 1.0 / d1;
      const double c21 = Quu[9] - l20 * Quu[4], c31 = Quu[13] - l30 * Quu[4];
      const double l21 = c21 * i1, l31 = c31 * i1;
      const double d2 = Quu[10] - l20 * Quu[8] - l21 * c21;
      const double i2 = 1.0 / d2;
      const double c32 = Quu[14] - l30 * Quu[8] - l31 * c21;
      const double l32 = c32 * i2;
      const double d3 = Quu[15] - l30 * Quu[12] - l31 * c31 - l32 * c32;
      const double i3 = 1.0 / d3;
      const double y0 = rhs[0], y1 = rhs[1] - l10 * y0, y2 = rhs[2] - l20 * y0 - l21 * y1,
                   y3 = rhs[3] - l30 * y0 - l31 * y1 - l32 * y2;
      const double x3 = y3 * i3, x2 = y2 * i2 - l32 * x3, x1 = y1 * i1 - l21 * x2 - l31 * x3,
                   x0 = y0 * i0 - l10 * x1 - l20 * x2 - l30 * x3;
      kcol_all[l][0] = -x0; kcol_all[l][1] = -x1; kcol_all[l][2] = -x2; kcol_all[l][3] = -x3;
      if (!sym) {  // the general kernel factors as the reference does: Eigen's diagonally pivoted LDL^T (backward_layout.h)
        double xs[4];
        ldlt4_pivoted_solve(Quu, rhs, xs);
        for (int aa = 0; aa < 4; ++aa) kcol_all[l][aa] = -xs[aa];
      }
      for (int bb = 0; bb < 4; ++bb)
        mc_all[l][bb] = kcol_all[l][0] * Quu[bb] + kcol_all[l][1] * Quu[4 + bb] + kcol_all[l][2] * Quu[8 + bb] +
                        kcol_all[l][3] * Quu[12 + bb];
      QuTk[l] += rhs[0] * kcol_all[l][0] + rhs[1] * kcol_all[l][1] + rhs[2] * kcol_all[l][2] + rhs[3] * kcol_all[l][3];
      kTQuuk[l] += mc_all[l][0] * kcol_all[l][0] + mc_all[l][1] * kcol_all[l][1] + mc_all[l][2] * kcol_all[l][2] +
                   mc_all[l][3] * kcol_all[l][3];
      (void)Quu_l;
      (void)kk;
    }
    for (int l = 0; l < 64; ++l) {
      const int j = l & 15, kk = l >> 4;
      const double *kcol = kcol_all[l], *mc = mc_all[l], *kff = kcol_all[12];  // k broadcast from lane 12
      if (sym) {
        vx[l] = ghat[l] + (kcol[0] * gs[12] + kcol[1] * gs[13] + kcol[2] * gs[14] + kcol[3] * gs[15]);
        aop[l] = H[l][3];
      } else {
        vx[l] = ghat[l] - (mc[0] * kff[0] + mc[1] * kff[1] + mc[2] * kff[2] + mc[3] * kff[3]);
        aop[l] = -mc[kk];
      }
      bop[l] = kcol[kk];
      if (kk == 0) {
        double *g = gains + (long)i * 52;
        if (j < 12) for (int aa = 0; aa < 4; ++aa) g[4 + 4 * j + aa] = kcol[aa];
        else if (j == 12) for (int aa = 0; aa < 4; ++aa) g[aa] = kcol[aa];
      }
    }
    mfma_f64_16x16x4(aop, bop, H);
    for (int l = 0; l < 64; ++l) {
      const int j = l & 15, kk = l >> 4;
      if (j < 12) {
        for (int r = 0; r < 3; ++r) Vs[(4 * r + kk) * LD + j] = H[l][r];
        if (kk == 0) vxs[j] = vx[l];
      }
    }
    for (int l = 0; l < 64; ++l) {
      const int j = l & 15, kk = l >> 4;
      for (int kc = 0; kc < 3; ++kc) {
        if (sym) va[l][kc] = H[l][kc];  // no transpose: V symmetric to rounding
        else va[l][kc] = (j < 12) ? Vs[j * LD + 4 * kc + kk] : 0.0;
        vxl[l][kc] = vxs[4 * kc + kk];
      }
    }
  }
  terms[0] = QuTk[12];  // lane 12 owns the feed-forward column
  terms[1] = sym ? -QuTk[12] : kTQuuk[12];
}

}  // extern "C"
