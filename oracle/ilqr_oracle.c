/*
 * ilqr_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ilqr_oracle.h).
 *
 * Dense, scalar restatement of the reference hot path.  No structure is
 * exploited on purpose: every 12x12 product the reference forms is formed here,
 * in the same association order, so that this file can be read side by side
 * with /root/reference/src/{ilqr.hh,cost.hh,quadrotor_model.cc}.
 *
 * Third-party arithmetic restated from the published algorithms:
 *   manif  @ ab560a3a1dac3f0f6bf5154056bb4408f4c9f67c  (SO3/SE3 exp, log, compose,
 *           inverse, adj, l/r Jacobians and inverses, Barfoot's Q block)
 *   Eigen  3.4.0 (quaternion -> matrix, 3x3 LLT, 4x4 pivoted LDLT + solve)
 * Items marked (+) are recollections of upstream thresholds that cannot be
 * checked offline; they act only when theta^2 <= 1e-10 or | |q|^2-1 | > 1e-10.
 */
#include "ilqr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#define MANIF_EPS 1e-10 /* manif Constants<double>::eps (+) */

/* ------------------------------------------------------------------ */
/* tiny dense helpers (row-major)                                      */
/* ------------------------------------------------------------------ */
static void mat_mul(const double *A, const double *B, double *C, int n, int k, int m) {
  /* C(n x m) = A(n x k) B(k x m) */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
      C[i * m + j] = s;
    }
}
static void mat_tmul(const double *A, const double *B, double *C, int n, int k, int m) {
  /* C(k x m) = A^T B with A(n x k), B(n x m) */
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0.0;
      for (int l = 0; l < n; ++l) s += A[l * k + i] * B[l * m + j];
      C[i * m + j] = s;
    }
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) { mat_mul(A, B, C, 3, 3, 3); }
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void mat3_T(const double A[9], double At[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) At[3 * i + j] = A[3 * j + i];
}
static void skew(const double a[3], double S[9]) {
  S[0] = 0;     S[1] = -a[2]; S[2] = a[1];
  S[3] = a[2];  S[4] = 0;     S[5] = -a[0];
  S[6] = -a[1]; S[7] = a[0];  S[8] = 0;
}
static void set_block(double *M, int ld, int r0, int c0, const double *B, int nr, int nc) {
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) M[(r0 + i) * ld + c0 + j] = B[i * nc + j];
}
static void identity(double *M, int n) {
  memset(M, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) M[i * n + i] = 1.0;
}

/* internal state: quaternion stored (x,y,z,w) as manif/Eigen coeffs() do */
typedef struct {
  double t[3];
  double q[4];
  double v[6];
} State;

static void unpack_state(const double x[13], State *s) {
  s->t[0] = x[0]; s->t[1] = x[1]; s->t[2] = x[2];
  s->q[3] = x[3]; s->q[0] = x[4]; s->q[1] = x[5]; s->q[2] = x[6]; /* wire order w,x,y,z */
  for (int i = 0; i < 6; ++i) s->v[i] = x[7 + i];
}
static void pack_state(const State *s, double x[13]) {
  x[0] = s->t[0]; x[1] = s->t[1]; x[2] = s->t[2];
  x[3] = s->q[3]; x[4] = s->q[0]; x[5] = s->q[1]; x[6] = s->q[2];
  for (int i = 0; i < 6; ++i) x[7 + i] = s->v[i];
}

/* Eigen 3.4.0 QuaternionBase::toRotationMatrix */
static void quat_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen quaternion product a*b */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

/* ------------------------------------------------------------------ */
/* manif SO3                                                           */
/* ------------------------------------------------------------------ */
/* SO3Tangent::exp: AngleAxis(theta, axis) -> quaternion; small angle [th/2, 1] (+) */
static void so3_exp(const double th[3], double q[4]) {
  const double th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  if (th2 > MANIF_EPS) {
    const double theta = sqrt(th2);
    /* Eigen normalized(): v / sqrt(squaredNorm) */
    const double ax = th[0] / theta, ay = th[1] / theta, az = th[2] / theta;
    const double ha = 0.5 * theta;
    const double s = sin(ha), c = cos(ha);
    q[3] = c; q[0] = s * ax; q[1] = s * ay; q[2] = s * az;
  } else {
    q[0] = th[0] / 2; q[1] = th[1] / 2; q[2] = th[2] / 2; q[3] = 1.0;
  }
}
/* SO3::log via atan2 on the quaternion, with the w<0 branch */
static void so3_log(const double q[4], double th[3]) {
  const double s2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  double log_coeff;
  if (s2 > MANIF_EPS) {
    const double s = sqrt(s2);
    const double c = q[3];
    const double two_angle = 2.0 * ((c < 0.0) ? atan2(-s, -c) : atan2(s, c));
    log_coeff = two_angle / s;
  } else {
    log_coeff = 2.0;
  }
  th[0] = q[0] * log_coeff; th[1] = q[1] * log_coeff; th[2] = q[2] * log_coeff;
}
/* SO3Tangent::ljac: I + (1-cos)/th^2 W + (th-sin)/th^3 W^2 */
static void so3_ljac(const double th[3], double J[9]) {
  const double th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  double W[9];
  skew(th, W);
  identity(J, 3);
  if (th2 <= MANIF_EPS) {
    for (int i = 0; i < 9; ++i) J[i] += 0.5 * W[i];
    return;
  }
  const double theta = sqrt(th2);
  double WW[9];
  mat3_mul(W, W, WW);
  const double a = (1.0 - cos(theta)) / th2;
  const double b = (theta - sin(theta)) / (th2 * theta);
  for (int i = 0; i < 9; ++i) J[i] += a * W[i] + b * WW[i];
}
/* SO3Tangent::ljacinv: I - W/2 + (1/th^2 - (1+cos)/(2 th sin)) W^2 */
static void so3_ljacinv(const double th[3], double J[9]) {
  const double th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  double W[9];
  skew(th, W);
  identity(J, 3);
  if (th2 <= MANIF_EPS) {
    for (int i = 0; i < 9; ++i) J[i] -= 0.5 * W[i];
    return;
  }
  const double theta = sqrt(th2);
  double WW[9];
  mat3_mul(W, W, WW);
  const double c = 1.0 / th2 - (1.0 + cos(theta)) / (2.0 * theta * sin(theta));
  for (int i = 0; i < 9; ++i) J[i] += -0.5 * W[i] + c * WW[i];
}
static void so3_rjac(const double th[3], double J[9]) {
  double L[9];
  so3_ljac(th, L);
  mat3_T(L, J);
}
static void so3_rjacinv(const double th[3], double J[9]) {
  double L[9];
  so3_ljacinv(th, L);
  mat3_T(L, J);
}

/* ------------------------------------------------------------------ */
/* manif SE3 (pose = t[3], q[4] xyzw)                                  */
/* ------------------------------------------------------------------ */
typedef struct {
  double t[3];
  double q[4];
} Pose;

/* SE3Tangent::exp: (ljac(theta) * rho, Exp(theta)) */
static void se3_exp(const double tau[6], Pose *T) {
  double Jl[9];
  so3_ljac(tau + 3, Jl);
  mat3_vec(Jl, tau, T->t);
  so3_exp(tau + 3, T->q);
}
/* SE3::log: theta = Log(R), rho = ljacinv(theta) * t */
static void se3_log(const Pose *T, double tau[6]) {
  double th[3], Ji[9];
  so3_log(T->q, th);
  so3_ljacinv(th, Ji);
  mat3_vec(Ji, T->t, tau);
  tau[3] = th[0]; tau[4] = th[1]; tau[5] = th[2];
}
/* SE3::compose: (R_a t_b + t_a, q_a * q_b), quaternion renormalised by the
 * first-order 2/(1+n) factor only when | |q|^2 - 1 | > eps (+) */
static void se3_compose(const Pose *A, const Pose *B, Pose *C) {
  double R[9], Rt[3], q[4];
  quat_to_R(A->q, R);
  mat3_vec(R, B->t, Rt);
  quat_mul(A->q, B->q, q);
  const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (fabs(n - 1.0) > MANIF_EPS) {
    const double scale = 2.0 / (1.0 + n);
    for (int i = 0; i < 4; ++i) q[i] *= scale;
  }
  for (int i = 0; i < 3; ++i) C->t[i] = Rt[i] + A->t[i];
  for (int i = 0; i < 4; ++i) C->q[i] = q[i];
}
/* SE3::inverse: (-R^T t, q^*) */
static void se3_inverse(const Pose *A, Pose *Ai) {
  double qc[4] = {-A->q[0], -A->q[1], -A->q[2], A->q[3]};
  double R[9], r[3];
  quat_to_R(qc, R);
  mat3_vec(R, A->t, r);
  for (int i = 0; i < 3; ++i) Ai->t[i] = -r[i];
  for (int i = 0; i < 4; ++i) Ai->q[i] = qc[i];
}
/* SE3::adj: [[R, hat(t) R],[0, R]] */
static void se3_adj(const Pose *T, double Ad[36]) {
  double R[9], S[9], SR[9], Z[9] = {0};
  quat_to_R(T->q, R);
  skew(T->t, S);
  mat3_mul(S, R, SR);
  set_block(Ad, 6, 0, 0, R, 3, 3);
  set_block(Ad, 6, 0, 3, SR, 3, 3);
  set_block(Ad, 6, 3, 0, Z, 3, 3);
  set_block(Ad, 6, 3, 3, R, 3, 3);
}
/* SE3Tangent::fillQ (Barfoot 2014, eq. 102) evaluated at c = [rho ; theta] */
static void se3_fillQ(const double c[6], double Qm[9]) {
  const double *rho = c, *th = c + 3;
  const double th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  const double A = 0.5;
  double B, C, D;
  if (th2 <= MANIF_EPS) {
    /* small-angle limits of the coefficients below (+) */
    B = 1. / 6. - th2 / 120.;
    C = -1. / 24. + th2 / 720.;
    D = -1. / 120.;
  } else {
    const double theta = sqrt(th2);
    const double s = sin(theta), co = cos(theta);
    B = (theta - s) / (th2 * theta);
    C = (1.0 - th2 / 2.0 - co) / (th2 * th2);
    D = 0.5 * (C - 3.0 * (theta - s - th2 * theta / 6.0) / (th2 * th2 * theta));
  }
  double V[9], W[9], VW[9], WV[9], WVW[9], VWW[9], WWV[9], WVWW[9], WWVW[9], WW[9];
  skew(rho, V);
  skew(th, W);
  mat3_mul(V, W, VW);
  mat3_mul(W, V, WV);
  mat3_mul(WV, W, WVW);
  mat3_mul(VW, W, VWW);
  mat3_mul(W, W, WW);
  mat3_mul(WW, V, WWV);
  mat3_mul(WVW, W, WVWW);
  mat3_mul(WW, VW, WWVW);
  for (int i = 0; i < 9; ++i)
    Qm[i] = A * V[i] + B * (WV[i] + VW[i] + WVW[i]) - C * (WWV[i] + VWW[i] - 3.0 * WVW[i]) -
            D * (WVWW[i] + WWVW[i]);
}
/* SE3Tangent::rjac: [[Jr(th), Q(-rho,-th)],[0, Jr(th)]] */
static void se3_rjac(const double tau[6], double J[36]) {
  double Jr[9], Qm[9], neg[6], Z[9] = {0};
  so3_rjac(tau + 3, Jr);
  for (int i = 0; i < 6; ++i) neg[i] = -tau[i];
  se3_fillQ(neg, Qm);
  set_block(J, 6, 0, 0, Jr, 3, 3);
  set_block(J, 6, 0, 3, Qm, 3, 3);
  set_block(J, 6, 3, 0, Z, 3, 3);
  set_block(J, 6, 3, 3, Jr, 3, 3);
}
/* SE3Tangent::rjacinv: [[Jr^-1, -Jr^-1 Q(-rho,-th) Jr^-1],[0, Jr^-1]] */
static void se3_rjacinv(const double tau[6], double J[36]) {
  double Ji[9], Qm[9], neg[6], T1[9], T2[9], Z[9] = {0};
  so3_rjacinv(tau + 3, Ji);
  for (int i = 0; i < 6; ++i) neg[i] = -tau[i];
  se3_fillQ(neg, Qm);
  mat3_mul(Ji, Qm, T1);
  mat3_mul(T1, Ji, T2);
  for (int i = 0; i < 9; ++i) T2[i] = -T2[i];
  set_block(J, 6, 0, 0, Ji, 3, 3);
  set_block(J, 6, 0, 3, T2, 3, 3);
  set_block(J, 6, 3, 0, Z, 3, 3);
  set_block(J, 6, 3, 3, Ji, 3, 3);
}
/* ljacinv(tau) = rjacinv(-tau) */
static void se3_ljacinv(const double tau[6], double J[36]) {
  double neg[6];
  for (int i = 0; i < 6; ++i) neg[i] = -tau[i];
  se3_rjacinv(neg, J);
}
/* SE3::rplus  T (+) tau = T * Exp(tau);  d/dT = Ad(Exp(tau)^-1), d/dtau = rjac(tau) */
static void se3_rplus(const Pose *T, const double tau[6], Pose *out, double *J_T, double *J_tau) {
  Pose E;
  se3_exp(tau, &E);
  se3_compose(T, &E, out);
  if (J_T) {
    Pose Ei;
    se3_inverse(&E, &Ei);
    se3_adj(&Ei, J_T);
  }
  if (J_tau) se3_rjac(tau, J_tau);
}
/* SE3::rminus  Y (-) X = Log(X^-1 Y);  d/dY = rjacinv(tau), d/dX = -ljacinv(tau) */
static void se3_rminus(const Pose *Y, const Pose *X, double tau[6], double *J_Y, double *J_X) {
  Pose Xi, D;
  se3_inverse(X, &Xi);
  se3_compose(&Xi, Y, &D);
  se3_log(&D, tau);
  if (J_Y) se3_rjacinv(tau, J_Y);
  if (J_X) {
    se3_ljacinv(tau, J_X);
    for (int i = 0; i < 36; ++i) J_X[i] = -J_X[i];
  }
}

/* -------- exported Lie-group wrappers (wire order w,x,y,z) -------- */
static void pose_in(const double T[7], Pose *P) {
  P->t[0] = T[0]; P->t[1] = T[1]; P->t[2] = T[2];
  P->q[3] = T[3]; P->q[0] = T[4]; P->q[1] = T[5]; P->q[2] = T[6];
}
static void pose_out(const Pose *P, double T[7]) {
  T[0] = P->t[0]; T[1] = P->t[1]; T[2] = P->t[2];
  T[3] = P->q[3]; T[4] = P->q[0]; T[5] = P->q[1]; T[6] = P->q[2];
}
void orc_so3_exp(const double th[3], double q[4]) {
  double qi[4];
  so3_exp(th, qi);
  q[0] = qi[3]; q[1] = qi[0]; q[2] = qi[1]; q[3] = qi[2];
}
void orc_so3_log(const double q[4], double th[3]) {
  double qi[4] = {q[1], q[2], q[3], q[0]};
  so3_log(qi, th);
}
void orc_so3_ljac(const double th[3], double J[9]) { so3_ljac(th, J); }
void orc_so3_ljacinv(const double th[3], double J[9]) { so3_ljacinv(th, J); }
void orc_se3_exp(const double tau[6], double T[7]) {
  Pose P;
  se3_exp(tau, &P);
  pose_out(&P, T);
}
void orc_se3_log(const double T[7], double tau[6]) {
  Pose P;
  pose_in(T, &P);
  se3_log(&P, tau);
}
void orc_se3_compose(const double A[7], const double B[7], double C[7]) {
  Pose a, b, c;
  pose_in(A, &a);
  pose_in(B, &b);
  se3_compose(&a, &b, &c);
  pose_out(&c, C);
}
void orc_se3_inverse(const double A[7], double Ai[7]) {
  Pose a, b;
  pose_in(A, &a);
  se3_inverse(&a, &b);
  pose_out(&b, Ai);
}
void orc_se3_adj(const double T[7], double Ad[36]) {
  Pose P;
  pose_in(T, &P);
  se3_adj(&P, Ad);
}
void orc_se3_rjac(const double tau[6], double J[36]) { se3_rjac(tau, J); }
void orc_se3_rjacinv(const double tau[6], double J[36]) { se3_rjacinv(tau, J); }
void orc_se3_ljacinv(const double tau[6], double J[36]) { se3_ljacinv(tau, J); }

/* ------------------------------------------------------------------ */
/* QuadrotorModel (quadrotor_model.cc)                                 */
/* ------------------------------------------------------------------ */
typedef struct {
  orc_model_params p;
  double moment_arms[12]; /* 3 x 4, quadrotor_model.cc:15-18 */
  double L[9];            /* lower Cholesky factor of the inertia (Eigen LLT) */
} Model;

/* Eigen LLT (unblocked, lower): returns 0 on success, 1 on NumericalIssue */
static int llt3(const double A[9], double L[9]) {
  memset(L, 0, sizeof(double) * 9);
  for (int k = 0; k < 3; ++k) {
    double x = A[k * 3 + k];
    for (int j = 0; j < k; ++j) x -= L[k * 3 + j] * L[k * 3 + j];
    if (!(x > 0.0)) return 1;
    x = sqrt(x);
    L[k * 3 + k] = x;
    for (int i = k + 1; i < 3; ++i) {
      double s = A[i * 3 + k];
      for (int j = 0; j < k; ++j) s -= L[i * 3 + j] * L[k * 3 + j];
      L[i * 3 + k] = s / x;
    }
  }
  return 0;
}
/* LLT::solve for nrhs columns (B, X are 3 x nrhs row-major) */
static void llt3_solve(const double L[9], const double *B, int nrhs, double *X) {
  for (int c = 0; c < nrhs; ++c) {
    double y[3];
    for (int i = 0; i < 3; ++i) {
      double s = B[i * nrhs + c];
      for (int j = 0; j < i; ++j) s -= L[i * 3 + j] * y[j];
      y[i] = s / L[i * 3 + i];
    }
    for (int i = 2; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < 3; ++j) s -= L[j * 3 + i] * X[j * nrhs + c];
      X[i * nrhs + c] = s / L[i * 3 + i];
    }
  }
}

/* quadrotor_model.cc:6-25 */
static int model_init(const orc_model_params *mp, Model *m) {
  m->p = *mp;
  const double a = mp->arm_length_m, c = mp->torque_to_thrust_ratio_m;
  const double ma[12] = {0, -a, 0, a, a, 0.0, -a, 0.0, -c, c, -c, c};
  memcpy(m->moment_arms, ma, sizeof(ma));
  const int bad = llt3(mp->inertia, m->L);
  /* Eigen isApprox(I, I^T): |a-b|^2 <= prec^2 min(|a|^2,|b|^2), prec = 1e-12 */
  double d2 = 0, n2 = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double d = mp->inertia[i * 3 + j] - mp->inertia[j * 3 + i];
      d2 += d * d;
      n2 += mp->inertia[i * 3 + j] * mp->inertia[i * 3 + j];
    }
  if (bad || !(d2 <= 1e-24 * n2)) return ORC_ERR_BAD_INERTIA;
  return ORC_OK;
}
int orc_model_check(const orc_model_params *mp) {
  Model m;
  return model_init(mp, &m);
}

/* quadrotor_model.cc:65-122 */
static void continuous_dynamics(const Model *m, const State *x, const double u[4],
                                double xdot[12], double *Jx, double *Ju) {
  const double g = m->p.g_mpss, mass = m->p.mass_kg;
  const double *I = m->p.inertia;
  double R[9];
  quat_to_R(x->q, R);
  const double RTez[3] = {R[6], R[7], R[8]}; /* R^T e_z */
  /* :67 xdot.body_velocity = x.body_velocity */
  for (int i = 0; i < 6; ++i) xdot[i] = x->v[i];
  /* :68-72 lin acc = -g R^T e_z + u.sum() e_z / m   (no -omega x v term) */
  const double usum = ((u[0] + u[1]) + u[2]) + u[3];
  xdot[6] = -g * RTez[0] + usum * 0.0 / mass;
  xdot[7] = -g * RTez[1] + usum * 0.0 / mass;
  xdot[8] = -g * RTez[2] + usum * 1.0 / mass;
  /* :74 M = moment_arms * u */
  double M[3];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 4; ++j) s += m->moment_arms[i * 4 + j] * u[j];
    M[i] = s;
  }
  /* :76-78 ang acc = I^-1 (M - (hat(w) I) w) */
  const double *w = x->v + 3;
  double What[9], WI[9], WIw[3], rhs[3];
  skew(w, What);
  mat3_mul(What, I, WI);
  mat3_vec(WI, w, WIw);
  for (int i = 0; i < 3; ++i) rhs[i] = M[i] - WIw[i];
  llt3_solve(m->L, rhs, 1, xdot + 9);

  if (Jx) {
    memset(Jx, 0, sizeof(double) * 144);
    /* :84-85 d(pose rate)/d(velocity) = I6 */
    for (int i = 0; i < 6; ++i) Jx[i * 12 + 6 + i] = 1.0;
    /* :88-96 d(lin acc)/d(rot) = -g hat(R^T e_z) */
    double H[9];
    skew(RTez, H);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Jx[(6 + i) * 12 + 3 + j] = -g * H[i * 3 + j];
    /* :99-111 d(ang acc)/d(omega) = -I^-1 (hat(w) I - hat(I w)) */
    double Iw[3], IwH[9], Jd[9], S[9];
    mat3_vec(I, w, Iw);
    skew(Iw, IwH);
    for (int i = 0; i < 9; ++i) Jd[i] = WI[i] - IwH[i];
    llt3_solve(m->L, Jd, 3, S);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Jx[(9 + i) * 12 + 9 + j] = -S[i * 3 + j];
  }
  if (Ju) {
    memset(Ju, 0, sizeof(double) * 48);
    /* :115-116 row body_lin_vel[2] = 1/m */
    for (int j = 0; j < 4; ++j) Ju[8 * 4 + j] = 1.0 / mass;
    /* :118-119 rows body_ang_vel = I^-1 moment_arms */
    double S[12];
    llt3_solve(m->L, m->moment_arms, 4, S);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) Ju[(9 + i) * 4 + j] = S[i * 4 + j];
  }
}

/* quadrotor_model.cc:174-206: add / operator+(State, StateTangent) */
static void state_add(const State *x, const double tg[12], State *out, double *J_lhs,
                      double *J_rhs) {
  Pose T, To;
  memcpy(T.t, x->t, sizeof(T.t));
  memcpy(T.q, x->q, sizeof(T.q));
  double J1[36], J2[36];
  se3_rplus(&T, tg, &To, J_lhs ? J1 : 0, J_rhs ? J2 : 0);
  memcpy(out->t, To.t, sizeof(To.t));
  memcpy(out->q, To.q, sizeof(To.q));
  for (int i = 0; i < 6; ++i) out->v[i] = x->v[i] + tg[6 + i];
  if (J_lhs) {
    identity(J_lhs, 12);
    set_block(J_lhs, 12, 0, 0, J1, 6, 6);
  }
  if (J_rhs) {
    identity(J_rhs, 12);
    set_block(J_rhs, 12, 0, 0, J2, 6, 6);
  }
}
/* quadrotor_model.cc:215-250: minus / operator-(State, State) */
static void state_minus(const State *l, const State *r, double out[12], double *J_lhs,
                        double *J_rhs) {
  Pose Y, X;
  memcpy(Y.t, l->t, sizeof(Y.t));
  memcpy(Y.q, l->q, sizeof(Y.q));
  memcpy(X.t, r->t, sizeof(X.t));
  memcpy(X.q, r->q, sizeof(X.q));
  double J1[36], J2[36];
  se3_rminus(&Y, &X, out, J_lhs ? J1 : 0, J_rhs ? J2 : 0);
  for (int i = 0; i < 6; ++i) out[6 + i] = l->v[i] - r->v[i];
  if (J_lhs) {
    identity(J_lhs, 12);
    set_block(J_lhs, 12, 0, 0, J1, 6, 6);
  }
  if (J_rhs) {
    identity(J_rhs, 12);
    for (int i = 0; i < 144; ++i) J_rhs[i] = -J_rhs[i];
    set_block(J_rhs, 12, 0, 0, J2, 6, 6);
  }
}
/* quadrotor_model.cc:266-276 */
static void euler_step(const State *x, const double xdot[12], double dt, State *out,
                       double *J_lhs, double *J_rhs) {
  double tg[12];
  for (int i = 0; i < 12; ++i) tg[i] = dt * xdot[i];
  state_add(x, tg, out, J_lhs, J_rhs);
  if (J_rhs)
    for (int i = 0; i < 144; ++i) J_rhs[i] *= dt;
}
/* quadrotor_model.cc:33-49 */
static void discrete_dynamics(const Model *m, const State *x, const double u[4], double dt,
                              State *xn, double *Jx, double *Ju) {
  double xdot[12];
  if (Jx && Ju) {
    double Jcx[144], Jcu[48], El[144], Er[144], T[144];
    continuous_dynamics(m, x, u, xdot, Jcx, Jcu);
    euler_step(x, xdot, dt, xn, El, Er);
    /* :43-45 J_x = J_lhs + J_rhs * Jc_x ; J_u = J_rhs * Jc_u */
    mat_mul(Er, Jcx, T, 12, 12, 12);
    for (int i = 0; i < 144; ++i) Jx[i] = El[i] + T[i];
    mat_mul(Er, Jcu, Ju, 12, 12, 4);
  } else {
    continuous_dynamics(m, x, u, xdot, 0, 0);
    euler_step(x, xdot, dt, xn, 0, 0);
  }
}

/* EXTENSION, not in the reference's executed code: the classical Runge-Kutta step that quadrotor_model.cc:51-63 sketches in a
 * comment --
 *     coeffs {1/6, 2/6, 2/6, 1/6}, dt table {0, dt/2, dt/2, dt};  k = 0, x_dot = 0;
 *     for i in 0..3:  k = continuous_dynamics(euler_step(x, k, dt_table[i]), u);  x_dot += coeffs[i] k
 * -- followed by x_next = euler_step(x, x_dot, dt) as in :39.  Every stage restarts from x (the sketch's euler_step(x, k, .)),
 * which on SE(3) x R^6 is the Runge-Kutta-Munthe-Kaas form with the tangent taken at x.
 * Jacobians by the chain rule through the reference's own differentials (euler_step's J_x_lhs / J_x_rhs, :266-276, and
 * continuous_dynamics' J_x / J_u, :80-119), the way discrete_dynamics composes them at :43-45:
 *     x_i = x (+) h_i k_{i-1}:   A_i = dx_i/dx = El_i + Er_i K_{i-1},      B_i = dx_i/du = Er_i Ku_{i-1}
 *     k_i = f(x_i, u):           K_i = dk_i/dx = F_x(x_i) A_i,             Ku_i = F_x(x_i) B_i + F_u
 *     x_next = x (+) dt sum c_i k_i:   J_x = El + Er sum c_i K_i,           J_u = Er sum c_i Ku_i                       */
static void discrete_dynamics_rk4(const Model *m, const State *x, const double u[4], double dt,
                                  State *xn, double *Jx, double *Ju) {
  const double coeffs[4] = {1.0 / 6.0, 2.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0};
  const double dt_table[4] = {0.0, dt / 2.0, dt / 2.0, dt};
  const int diffs = (Jx && Ju);
  double k[12] = {0}, xdot[12] = {0};
  double K[144] = {0}, Ku[48] = {0}, SK[144] = {0}, SKu[48] = {0};
  for (int i = 0; i < 4; ++i) {
    State xi;
    double El[144], Er[144], Fx[144], Fu[48], A[144], B[48], T[144];
    euler_step(x, k, dt_table[i], &xi, diffs ? El : 0, diffs ? Er : 0);
    if (diffs) {
      mat_mul(Er, K, T, 12, 12, 12);
      for (int e = 0; e < 144; ++e) A[e] = El[e] + T[e];
      mat_mul(Er, Ku, B, 12, 12, 4);
    }
    continuous_dynamics(m, &xi, u, k, diffs ? Fx : 0, diffs ? Fu : 0);
    if (diffs) {
      mat_mul(Fx, A, K, 12, 12, 12);
      mat_mul(Fx, B, Ku, 12, 12, 4);
      for (int e = 0; e < 48; ++e) Ku[e] += Fu[e];
      for (int e = 0; e < 144; ++e) SK[e] += coeffs[i] * K[e];
      for (int e = 0; e < 48; ++e) SKu[e] += coeffs[i] * Ku[e];
    }
    for (int e = 0; e < 12; ++e) xdot[e] += coeffs[i] * k[e];
  }
  if (diffs) {
    double El[144], Er[144], T[144];
    euler_step(x, xdot, dt, xn, El, Er);
    mat_mul(Er, SK, T, 12, 12, 12);
    for (int e = 0; e < 144; ++e) Jx[e] = El[e] + T[e];
    mat_mul(Er, SKu, Ju, 12, 12, 4);
  } else {
    euler_step(x, xdot, dt, xn, 0, 0);
  }
}
/* the step the solver integrates with: 0 = the reference's explicit Euler, 1 = the sketched Runge-Kutta (extension) */
static void discrete_step(const Model *m, int integrator, const State *x, const double u[4], double dt,
                          State *xn, double *Jx, double *Ju) {
  if (integrator == 1) discrete_dynamics_rk4(m, x, u, dt, xn, Jx, Ju);
  else discrete_dynamics(m, x, u, dt, xn, Jx, Ju);
}

int orc_continuous_dynamics(const orc_model_params *mp, const double x[13], const double u[4],
                            double xdot[12], double *Jx, double *Ju) {
  Model m;
  const int rc = model_init(mp, &m);
  if (rc) return rc;
  State s;
  unpack_state(x, &s);
  continuous_dynamics(&m, &s, u, xdot, Jx, Ju);
  return ORC_OK;
}
int orc_discrete_dynamics(const orc_model_params *mp, const double x[13], const double u[4],
                          double dt, double xnext[13], double *Jx, double *Ju) {
  return orc_discrete_step(mp, 0, x, u, dt, xnext, Jx, Ju);
}
int orc_discrete_step(const orc_model_params *mp, int integrator, const double x[13], const double u[4],
                      double dt, double xnext[13], double *Jx, double *Ju) {
  Model m;
  const int rc = model_init(mp, &m);
  if (rc) return rc;
  if (integrator != 0 && integrator != 1) return ORC_ERR_INVALID;
  State s, n;
  unpack_state(x, &s);
  double jx[144], ju[48];
  const int want = (Jx || Ju);
  discrete_step(&m, integrator, &s, u, dt, &n, want ? jx : 0, want ? ju : 0);
  if (Jx) memcpy(Jx, jx, sizeof(jx));
  if (Ju) memcpy(Ju, ju, sizeof(ju));
  pack_state(&n, xnext);
  return ORC_OK;
}
void orc_state_add(const double x[13], const double tangent[12], double out[13], double *J_lhs,
                   double *J_rhs) {
  State s, o;
  unpack_state(x, &s);
  state_add(&s, tangent, &o, J_lhs, J_rhs);
  pack_state(&o, out);
}
void orc_state_minus(const double lhs[13], const double rhs[13], double out[12], double *J_lhs,
                     double *J_rhs) {
  State l, r;
  unpack_state(lhs, &l);
  unpack_state(rhs, &r);
  state_minus(&l, &r, out, J_lhs, J_rhs);
}
void orc_euler_step(const double x[13], const double xdot[12], double dt, double out[13],
                    double *J_lhs, double *J_rhs) {
  State s, o;
  unpack_state(x, &s);
  euler_step(&s, xdot, dt, &o, J_lhs, J_rhs);
  pack_state(&o, out);
}

/* ------------------------------------------------------------------ */
/* CostFunction::operator() (cost.hh:36-61)                            */
/* ------------------------------------------------------------------ */
static double cost_fn(const double Q[144], const double R[16], const State *x, const double u[4],
                      const State *xd, const double ud[4], double *Cx, double *Cu, double *Cxx,
                      double *Cuu, double *Cxu) {
  double dx[12], J[144], du[4];
  /* :42-43 the reference always evaluates the (-) Jacobians; the value is the same
   * either way, so the oracle only forms them when differentials are requested */
  state_minus(x, xd, dx, Cx ? J : 0, 0);
  for (int i = 0; i < 4; ++i) du[i] = u[i] - ud[i];
  /* :47-48 (dx^T Q) dx + (du^T R) du */
  double xq[12], ur[4];
  mat_tmul(dx, Q, xq, 12, 1, 12);
  mat_tmul(du, R, ur, 4, 1, 4);
  double cx = 0, cu = 0;
  for (int i = 0; i < 12; ++i) cx += xq[i] * dx[i];
  for (int i = 0; i < 4; ++i) cu += ur[i] * du[i];
  const double cost = cx + cu;
  if (Cx) {
    /* :51 x = ((2 dx^T) Q) J */
    double dx2[12], t[12];
    for (int i = 0; i < 12; ++i) dx2[i] = 2 * dx[i];
    mat_tmul(dx2, Q, t, 12, 1, 12);
    mat_mul(t, J, Cx, 1, 12, 12);
    /* :52 xx = ((2 J^T) Q) J */
    double Jt2[144], T[144];
    for (int i = 0; i < 12; ++i)
      for (int j = 0; j < 12; ++j) Jt2[i * 12 + j] = 2 * J[j * 12 + i];
    mat_mul(Jt2, Q, T, 12, 12, 12);
    mat_mul(T, J, Cxx, 12, 12, 12);
    /* :54-55 u = (2 du^T) R ; uu = 2 R */
    double du2[4];
    for (int i = 0; i < 4; ++i) du2[i] = 2 * du[i];
    mat_tmul(du2, R, Cu, 4, 1, 4);
    for (int i = 0; i < 16; ++i) Cuu[i] = 2 * R[i];
    /* :57 xu = 0 */
    memset(Cxu, 0, sizeof(double) * 48);
  }
  return cost;
}
double orc_cost(const double Q[144], const double R[16], const double x[13], const double u[4],
                const double xd[13], const double ud[4], double *Cx, double *Cu, double *Cxx,
                double *Cuu, double *Cxu) {
  State s, d;
  unpack_state(x, &s);
  unpack_state(xd, &d);
  double cx[12], cu[4], cxx[144], cuu[16], cxu[48];
  const int want = Cx || Cu || Cxx || Cuu || Cxu;
  const double c = cost_fn(Q, R, &s, u, &d, ud, want ? cx : 0, cu, cxx, cuu, cxu);
  if (Cx) memcpy(Cx, cx, sizeof(cx));
  if (Cu) memcpy(Cu, cu, sizeof(cu));
  if (Cxx) memcpy(Cxx, cxx, sizeof(cxx));
  if (Cuu) memcpy(Cuu, cuu, sizeof(cuu));
  if (Cxu) memcpy(Cxu, cxu, sizeof(cxu));
  return c;
}

/* ------------------------------------------------------------------ */
/* Eigen 3.4.0 LDLT<Matrix4d, Lower>: in-place, diagonal pivoting       */
/* ------------------------------------------------------------------ */
void orc_ldlt4_solve(const double A[16], const double *B, int nrhs, double *X) {
  enum { n = 4 };
  double m[16];
  int tr[n];
  memcpy(m, A, sizeof(m)); /* only the lower triangle is read */
  for (int k = 0; k < n; ++k) {
    /* pivot: biggest |diagonal| in the trailing corner (first one on ties) */
    int big = k;
    double best = fabs(m[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(m[i * n + i]) > best) {
        best = fabs(m[i * n + i]);
        big = i;
      }
    tr[k] = big;
    if (k != big) {
      /* symmetric row/column swap restricted to the lower triangle */
      for (int j = 0; j < k; ++j) {
        const double t = m[k * n + j];
        m[k * n + j] = m[big * n + j];
        m[big * n + j] = t;
      }
      for (int i = big + 1; i < n; ++i) {
        const double t = m[i * n + k];
        m[i * n + k] = m[i * n + big];
        m[i * n + big] = t;
      }
      {
        const double t = m[k * n + k];
        m[k * n + k] = m[big * n + big];
        m[big * n + big] = t;
      }
      for (int i = k + 1; i < big; ++i) {
        const double t = m[i * n + k];
        m[i * n + k] = m[big * n + i];
        m[big * n + i] = t;
      }
    }
    /* A10 = row k head k ; A20 = block below ; A21 = column k tail */
    if (k > 0) {
      double temp[n];
      for (int j = 0; j < k; ++j) temp[j] = m[j * n + j] * m[k * n + j];
      double s = 0;
      for (int j = 0; j < k; ++j) s += m[k * n + j] * temp[j];
      m[k * n + k] -= s;
      for (int i = k + 1; i < n; ++i) {
        double r = 0;
        for (int j = 0; j < k; ++j) r += m[i * n + j] * temp[j];
        m[i * n + k] -= r;
      }
    }
    const double akk = m[k * n + k];
    if (fabs(akk) > 0.0)
      for (int i = k + 1; i < n; ++i) m[i * n + k] /= akk;
  }
  /* solve: x = P^T L^-T D^-1 L^-1 P b */
  const double tol = 2.2250738585072014e-308; /* numeric_limits<double>::min() */
  for (int c = 0; c < nrhs; ++c) {
    double y[n];
    for (int i = 0; i < n; ++i) y[i] = B[i * nrhs + c];
    for (int k = 0; k < n; ++k)
      if (tr[k] != k) {
        const double t = y[k];
        y[k] = y[tr[k]];
        y[tr[k]] = t;
      }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) y[i] -= m[i * n + j] * y[j];
    for (int i = 0; i < n; ++i) y[i] = (fabs(m[i * n + i]) > tol) ? y[i] / m[i * n + i] : 0.0;
    for (int i = n - 1; i >= 0; --i)
      for (int j = i + 1; j < n; ++j) y[i] -= m[j * n + i] * y[j];
    for (int k = n - 1; k >= 0; --k)
      if (tr[k] != k) {
        const double t = y[k];
        y[k] = y[tr[k]];
        y[tr[k]] = t;
      }
    for (int i = 0; i < n; ++i) X[i * nrhs + c] = y[i];
  }
}

/* ------------------------------------------------------------------ */
/* ILQR (ilqr.hh)                                                      */
/* ------------------------------------------------------------------ */
struct orc_solver {
  Model model;
  double Q[144], R[16];
  double *desired; /* n_desired x 18 */
  int n_desired;
  double dt;
  orc_options opt;
  /* Levenberg-Marquardt restarts: NOT in the reference (extension); all zero = off */
  double mu_init, mu_factor, mu_max;
  /* integrator of the discrete dynamics: 0 = explicit Euler (the reference), 1 = Runge-Kutta (extension, see discrete_dynamics_rk4) */
  int integrator;
  /* form of the value recursion: 0 = ilqr.hh:132-133 as written (the reference), 1 = the substituted, symmetrised form (EXTENSION, see
   * orc_set_recursion) */
  int recursion;
};

int orc_solver_create(const orc_model_params *mp, const double Q[144], const double R[16],
                      const double *desired, int n_desired, double dt, const orc_options *opt,
                      orc_solver **out) {
  if (!mp || !Q || !R || !opt || !out || n_desired < 0 || (n_desired > 0 && !desired))
    return ORC_ERR_INVALID;
  orc_solver *s = (orc_solver *)calloc(1, sizeof(orc_solver));
  const int rc = model_init(mp, &s->model);
  if (rc) {
    free(s);
    return rc;
  }
  memcpy(s->Q, Q, sizeof(s->Q));
  memcpy(s->R, R, sizeof(s->R));
  s->n_desired = n_desired;
  s->desired = (double *)malloc(sizeof(double) * ORC_PT * (n_desired > 0 ? n_desired : 1));
  if (n_desired > 0) memcpy(s->desired, desired, sizeof(double) * ORC_PT * n_desired);
  s->dt = dt;
  s->opt = *opt;
  *out = s;
  return ORC_OK;
}
void orc_solver_destroy(orc_solver *s) {
  if (!s) return;
  free(s->desired);
  free(s);
}

/* ilqr.hh:13-22 */
static double cost_reduction(const double terms[2], double step) {
  return step * terms[0] + step * step * terms[1] / 2.0;
}
/* ilqr.hh:196-205 */
static int is_converged(const orc_options *o, double cost, double new_cost) {
  if (fabs(cost - new_cost) / fabs(cost) < o->rtol) return 1;
  if (fabs(cost - new_cost) < o->atol) return 1;
  return 0;
}

/* ---- the comparisons a solve's control flow depends on, with their margins (iteration and trial counts
 * "must match except where the deciding margin is < 1e-9 relative (log those cases)").  A margin is the distance of the compared
 * quantity from the value at which the comparison flips, relative to the iteration's cost -- the scale on which two
 * implementations of the same arithmetic differ (their costs agree to about 1e-13 relative, so a comparison of cost differences
 * can only come out differently when it is that close to flipping).  Recording changes nothing in the solve. */
typedef struct {
  orc_decision *dec;
  int cap, n;
  int n_bwd, n_fwd; /* passes completed when the comparison is made */
} DecLog;
static void dec_push(DecLog *log, int kind, int iter, int trial, int result, double lhs, double rhs, double margin) {
  if (!log) return;
  if (log->dec && log->n < log->cap) {
    orc_decision *d = &log->dec[log->n];
    d->kind = kind; d->iter = iter; d->trial = trial; d->result = result;
    d->lhs = lhs; d->rhs = rhs; d->margin = margin;
    d->n_bwd = log->n_bwd; d->n_fwd = log->n_fwd;
  }
  ++log->n;
}
/* margin of is_converged(cost, new_cost): converged -> the larger slack of the tests that hold; not converged -> the smaller
 * excess of the two (both fail), each as a cost difference over |cost| */
static double converged_margin(const orc_options *o, double cost, double new_cost) {
  const double d = fabs(cost - new_cost), c = fabs(cost);
  const double sa = o->rtol * c - d, sb = o->atol - d; /* slack of the relative / the absolute test (positive: holds) */
  const int a = d / c < o->rtol, b = d < o->atol;
  double m;
  if (a || b) m = (a && b) ? (sa > sb ? sa : sb) : (a ? sa : sb);
  else m = (-sa < -sb) ? -sa : -sb;
  return c > 0.0 ? fabs(m) / c : fabs(m);
}

static double knot_cost(const orc_solver *s, const double *pt, int i, double *Cx, double *Cu,
                        double *Cxx, double *Cuu, double *Cxu) {
  State x, xd;
  unpack_state(pt + 1, &x);
  unpack_state(s->desired + i * ORC_PT + 1, &xd);
  return cost_fn(s->Q, s->R, &x, pt + 14, &xd, s->desired + i * ORC_PT + 14, Cx, Cu, Cxx, Cuu,
                 Cxu);
}

/* ilqr.hh:89-95 */
int orc_cost_trajectory(const orc_solver *s, const double *traj, int n, double *cost) {
  if (n > s->n_desired) return ORC_ERR_LENGTH_MISMATCH;
  double c = 0.0;
  for (int i = 0; i < n; ++i) c += knot_cost(s, traj + i * ORC_PT, i, 0, 0, 0, 0, 0);
  *cost = c;
  return ORC_OK;
}

/* Extension (not in the reference): switch Levenberg-Marquardt restarts on (mu_init > 0) or off. */
int orc_set_regularisation(orc_solver *s, double mu_init, double mu_factor, double mu_max) {
  if (!s || !(mu_init >= 0.0)) return ORC_ERR_INVALID;
  if (mu_init > 0.0 && (!(mu_factor > 1.0) || !(mu_max >= mu_init))) return ORC_ERR_INVALID;
  s->mu_init = mu_init;
  s->mu_factor = mu_init > 0.0 ? mu_factor : 1.0;
  s->mu_max = mu_init > 0.0 ? mu_max : 0.0;
  return ORC_OK;
}

int orc_set_integrator(orc_solver *s, int integrator) {
  if (!s || (integrator != 0 && integrator != 1)) return ORC_ERR_INVALID;
  s->integrator = integrator;
  return ORC_OK;
}
/* EXTENSION (default 0 = the reference).  mode 1 evaluates the value update of ilqr.hh:132-133 and the term of :139 with
 * K = -Q_uu^-1 Q_ux and k = -Q_uu^-1 Q_u substituted (exact for the exact solve with symmetric Q_uu) and symmetrises V_xx:
 *     V_x  = Q_x  + K^T Q_u            (= Q_x  - K^T Q_uu k)
 *     V_xx = Q_xx + Q_xu K             (= Q_xx - K^T Q_uu K),   then V_xx <- (V_xx + V_xx^T) / 2
 *     k^T Q_uu k = -(Q_u^T k)
 * Everything else (Q assembly :118-124, the pivoted LDL^T solves :126-128, Q_u^T k :138) is the reference's.  Why it exists: the
 * reference's unsymmetrised V_xx amplifies its rounding asymmetry from knot to knot and is noise beyond about 150 knots (DESIGN.md
 * section 4, finding), so at the horizons of BASELINE.json configs[2] (200 knots) and configs[4] (500 knots) the reference recursion
 * has no usable answer; this form is the one the symmetric-weight device kernels document, stated independently of them in dense
 * scalar C so that those configurations have a comparand outside the library.  tests/test_oracle_recursion.py holds it to the
 * reference form (gains 1e-10 relative) wherever that one is stable. */
int orc_set_recursion(orc_solver *s, int mode) {
  if (!s || (mode != 0 && mode != 1)) return ORC_ERR_INVALID;
  s->recursion = mode;
  return ORC_OK;
}
int orc_backwards_pass(const orc_solver *s, const double *traj, int n, double *gains,
                       double terms[2]) {
  return orc_backwards_pass_reg(s, traj, n, 0.0, gains, terms);
}

/* ilqr.hh:97-147; mu (extension, 0 in the reference) is added to the diagonal of Q_uu before it is
 * used anywhere */
int orc_backwards_pass_reg(const orc_solver *s, const double *traj, int n, double mu, double *gains,
                           double terms[2]) {
  if (n > s->n_desired) return ORC_ERR_LENGTH_MISMATCH;
  double v_x[12] = {0}, v_xx[144] = {0};
  terms[0] = 0.0;
  terms[1] = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    const double *pt = traj + i * ORC_PT;
    State x, xn;
    unpack_state(pt + 1, &x);
    double Jx[144], Ju[48];
    discrete_step(&s->model, s->integrator, &x, pt + 14, s->dt, &xn, Jx, Ju); /* :111-113 */
    double Cx[12], Cu[4], Cxx[144], Cuu[16], Cxu[48];
    knot_cost(s, pt, i, Cx, Cu, Cxx, Cuu, Cxu); /* :115-116 */

    /* :118-124 */
    double Qx[12], Qu[4], Qxx[144], Quu[16], Qxu[48];
    double t12[12], t4[4], JxTV[144], T[144], JuTV[48], T16[16], T48[48];
    mat_tmul(Jx, v_x, t12, 12, 12, 1);
    for (int a = 0; a < 12; ++a) Qx[a] = Cx[a] + t12[a];
    mat_tmul(Ju, v_x, t4, 12, 4, 1);
    for (int a = 0; a < 4; ++a) Qu[a] = Cu[a] + t4[a];
    mat_tmul(Jx, v_xx, JxTV, 12, 12, 12);
    mat_mul(JxTV, Jx, T, 12, 12, 12);
    for (int a = 0; a < 144; ++a) Qxx[a] = Cxx[a] + T[a];
    mat_tmul(Ju, v_xx, JuTV, 12, 4, 12);
    mat_mul(JuTV, Ju, T16, 4, 12, 4);
    for (int a = 0; a < 16; ++a) Quu[a] = Cuu[a] + T16[a];
    if (mu != 0.0)
      for (int a = 0; a < 4; ++a) Quu[a * 5] += mu;
    mat_mul(JxTV, Ju, T48, 12, 12, 4);
    for (int a = 0; a < 48; ++a) Qxu[a] = Cxu[a] + T48[a];

    /* :126-128 K = -ldlt(Quu).solve(Qxu^T) ; k = -ldlt(Quu).solve(Qu) */
    double QxuT[48], K[48], k[4];
    for (int a = 0; a < 12; ++a)
      for (int b = 0; b < 4; ++b) QxuT[b * 12 + a] = Qxu[a * 4 + b];
    orc_ldlt4_solve(Quu, QxuT, 12, K);
    orc_ldlt4_solve(Quu, Qu, 1, k);
    for (int a = 0; a < 48; ++a) K[a] = -K[a];
    for (int a = 0; a < 4; ++a) k[a] = -k[a];

    double *g = gains + (size_t)i * ORC_GAIN; /* written at i == the std::reverse at :143 */
    for (int a = 0; a < 4; ++a) g[a] = k[a];
    for (int c = 0; c < 12; ++c)
      for (int a = 0; a < 4; ++a) g[4 + c * 4 + a] = K[a * 12 + c];

    double quk = 0;
    for (int a = 0; a < 4; ++a) quk += Qu[a] * k[a];
    if (s->recursion == 1) {
      /* EXTENSION (orc_set_recursion): v_x = Qx + K^T Qu ; v_xx = sym(Qxx + Qxu K) ; k^T Quu k = -Qu^T k */
      double t12b[12], T2[144];
      mat_tmul(K, Qu, t12b, 4, 12, 1);
      for (int a = 0; a < 12; ++a) v_x[a] = Qx[a] + t12b[a];
      mat_mul(Qxu, K, T2, 12, 4, 12);
      for (int a = 0; a < 144; ++a) T2[a] = Qxx[a] + T2[a];
      for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) v_xx[a * 12 + b] = 0.5 * (T2[a * 12 + b] + T2[b * 12 + a]);
      terms[0] += quk;
      terms[1] += -quk;
      continue;
    }
    /* :132-133 v_x = Qx - (K^T Quu) k ; v_xx = Qxx - (K^T Quu) K */
    double KtQ[48], t12b[12], T2[144];
    mat_tmul(K, Quu, KtQ, 4, 12, 4);
    mat_mul(KtQ, k, t12b, 12, 4, 1);
    for (int a = 0; a < 12; ++a) v_x[a] = Qx[a] - t12b[a];
    mat_mul(KtQ, K, T2, 12, 4, 12);
    for (int a = 0; a < 144; ++a) v_xx[a] = Qxx[a] - T2[a];

    /* :136-140 */
    terms[0] += quk;
    double kq[4], kqk = 0;
    mat_tmul(k, Quu, kq, 4, 1, 4);
    for (int a = 0; a < 4; ++a) kqk += kq[a] * k[a];
    terms[1] += kqk;
  }
  return ORC_OK;
}

/* ilqr.hh:149-172 */
int orc_forward_sim(const orc_solver *s, const double *traj, int n, const double *gains,
                    double alpha, double *out) {
  if (n <= 0) return ORC_ERR_INVALID; /* the reference calls front() on an empty vector */
  State state;
  unpack_state(traj + 1, &state);
  for (int i = 0; i < n; ++i) {
    const double *pt = traj + i * ORC_PT;
    const double *g = gains + (size_t)i * ORC_GAIN;
    State xi;
    unpack_state(pt + 1, &xi);
    double dx[12];
    state_minus(&state, &xi, dx, 0, 0);
    double u[4];
    for (int a = 0; a < 4; ++a) {
      /* (u_i + alpha k) + K dx, K dx accumulated column by column (col-major gemv) */
      double kd = 0;
      for (int c = 0; c < 12; ++c) kd += g[4 + c * 4 + a] * dx[c];
      u[a] = (pt[14 + a] + alpha * g[a]) + kd;
    }
    double *o = out + i * ORC_PT;
    o[0] = pt[0];
    pack_state(&state, o + 1);
    for (int a = 0; a < 4; ++a) o[14 + a] = u[a];
    State nx;
    discrete_step(&s->model, s->integrator, &state, u, s->dt, &nx, 0, 0); /* :168, last one discarded */
    state = nx;
  }
  return ORC_OK;
}

/* ilqr.hh:174-194 */
static int line_search_logged(const orc_solver *s, const double *traj, int n, double cost,
                    const double *gains, const double terms[2], double *out_traj,
                    double *out_cost, double *out_step, int *out_trials, DecLog *log, int iter) {
  double step = 1.0;
  for (int i = 0; i < s->opt.ls_max_iters; ++i) {
    orc_forward_sim(s, traj, n, gains, step, out_traj);
    double new_cost;
    const int rc = orc_cost_trajectory(s, out_traj, n, &new_cost);
    if (rc) return -rc;
    const double desired = s->opt.desired_reduction_frac * cost_reduction(terms, step);
    if (log) ++log->n_fwd;
    dec_push(log, ORC_DEC_ARMIJO, iter, i, new_cost - cost < desired, new_cost - cost, desired,
             fabs((new_cost - cost) - desired) / (fabs(cost) > 0.0 ? fabs(cost) : 1.0));
    if (new_cost - cost < desired) {
      *out_cost = new_cost;
      *out_step = step;
      if (out_trials) *out_trials = i + 1;
      return ORC_STATUS_CONVERGED_EXPECTED; /* 0 = accepted */
    }
    step *= s->opt.step_update;
  }
  if (out_trials) *out_trials = s->opt.ls_max_iters;
  return ORC_STATUS_LINE_SEARCH_FAILED;
}
int orc_line_search(const orc_solver *s, const double *traj, int n, double cost,
                    const double *gains, const double terms[2], double *out_traj,
                    double *out_cost, double *out_step, int *out_trials) {
  return line_search_logged(s, traj, n, cost, gains, terms, out_traj, out_cost, out_step, out_trials, 0, 0);
}

/* ilqr.hh:53-87 */
static int solve_logged(const orc_solver *s, const double *init, int n, double *out_traj, double *out_cost,
              int *out_status, int *out_iters, int *out_n_bwd, int *out_n_fwd,
              double *cost_hist, double *debug_trajs, int cap, int *out_n_hist, DecLog *log) {
  if (n <= 0) return ORC_ERR_INVALID;
  if (n > s->n_desired) return ORC_ERR_LENGTH_MISMATCH;
  const size_t tsz = sizeof(double) * ORC_PT * (size_t)n;
  double *traj = (double *)malloc(tsz);
  double *cand = (double *)malloc(tsz);
  double *gains = (double *)malloc(sizeof(double) * ORC_GAIN * (size_t)n);
  memcpy(traj, init, tsz);
  double new_cost;
  orc_cost_trajectory(s, traj, n, &new_cost);
  int status = ORC_STATUS_MAX_ITERS, iters = 0, n_bwd = 0, n_fwd = 0, n_hist = 0;
  double mu = 0.0; /* extension; stays 0 (the reference) unless restarts are switched on */
  for (int i = 0; i < s->opt.max_iters;) {
    double terms[2];
    orc_backwards_pass_reg(s, traj, n, mu, gains, terms);
    ++n_bwd;
    if (log) { log->n_bwd = n_bwd; log->n_fwd = n_fwd; }
    const double cost = new_cost;
    const double expected_new_cost = cost + cost_reduction(terms, 1.0);
    if (i > 0)
      dec_push(log, ORC_DEC_EXPECTED, i, 0, is_converged(&s->opt, cost, expected_new_cost), cost, expected_new_cost,
               converged_margin(&s->opt, cost, expected_new_cost));
    if (i > 0 && is_converged(&s->opt, cost, expected_new_cost)) {
      status = ORC_STATUS_CONVERGED_EXPECTED;
      break;
    }
    if (i == 0) {
      orc_forward_sim(s, traj, n, gains, 1.0, cand);
      orc_cost_trajectory(s, cand, n, &new_cost);
      ++n_fwd;
      double *t = traj; traj = cand; cand = t;
    } else {
      double step;
      int trials = 0;
      const int ls = line_search_logged(s, traj, n, cost, gains, terms, cand, &new_cost, &step,
                                        &trials, log, i);
      n_fwd += trials;
      if (ls == ORC_STATUS_LINE_SEARCH_FAILED) {
        new_cost = cost;
        if (s->mu_init > 0.0) {
          /* extension: same iterate, larger mu, backward pass again (not an iteration) */
          const double next = (mu > 0.0) ? mu * s->mu_factor : s->mu_init;
          if (next <= s->mu_max) {
            mu = next;
            continue;
          }
        }
        status = ORC_STATUS_LINE_SEARCH_FAILED;
        break;
      }
      double *t = traj; traj = cand; cand = t;
    }
    if (mu > 0.0) {
      mu = mu / s->mu_factor;
      if (mu < s->mu_init) mu = 0.0;
    }
    ++iters;
    if (cost_hist && n_hist < cap) {
      cost_hist[n_hist] = new_cost;
      if (debug_trajs) memcpy(debug_trajs + (size_t)n_hist * ORC_PT * n, traj, tsz);
    }
    ++n_hist;
    if (log) log->n_fwd = n_fwd;
    if (i > 0)
      dec_push(log, ORC_DEC_CONVERGED, i, 0, is_converged(&s->opt, cost, new_cost), cost, new_cost,
               converged_margin(&s->opt, cost, new_cost));
    if (i > 0 && is_converged(&s->opt, cost, new_cost)) {
      status = ORC_STATUS_CONVERGED;
      break;
    }
    ++i;
  }
  memcpy(out_traj, traj, tsz);
  if (out_cost) *out_cost = new_cost;
  if (out_status) *out_status = status;
  if (out_iters) *out_iters = iters;
  if (out_n_bwd) *out_n_bwd = n_bwd;
  if (out_n_fwd) *out_n_fwd = n_fwd;
  if (out_n_hist) *out_n_hist = n_hist;
  free(traj);
  free(cand);
  free(gains);
  return ORC_OK;
}

int orc_solve(const orc_solver *s, const double *init, int n, double *out_traj, double *out_cost,
              int *out_status, int *out_iters, int *out_n_bwd, int *out_n_fwd,
              double *cost_hist, double *debug_trajs, int cap, int *out_n_hist) {
  return solve_logged(s, init, n, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd, cost_hist, debug_trajs, cap,
                      out_n_hist, 0);
}
/* the same solve, every comparison its control flow took recorded with its margin (dec[dec_cap]; *n_dec = how many there were) */
int orc_solve_decisions(const orc_solver *s, const double *init, int n, double *out_traj, double *out_cost,
                        int *out_status, int *out_iters, int *out_n_bwd, int *out_n_fwd, double *cost_hist, int cap,
                        int *out_n_hist, orc_decision *dec, int dec_cap, int *n_dec) {
  DecLog log = {dec, dec_cap, 0, 0, 0};
  const int rc = solve_logged(s, init, n, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd, cost_hist, 0, cap,
                              out_n_hist, &log);
  if (n_dec) *n_dec = log.n;
  return rc;
}

/* ------------------------------------------------------------------ */
/* threaded batch (cpu_baseline timing only)                           */
/* ------------------------------------------------------------------ */
typedef struct {
  const orc_solver *s;
  const double *init;
  int B, n;
  atomic_int *next; /* the next problem nobody has taken: threads draw problems one at a time (solve times differ by 3 x) */
  double *out_traj, *out_cost;
  int *out_status, *out_iters, *out_n_bwd, *out_n_fwd;
} BatchJob;

static void *batch_worker(void *arg) {
  BatchJob *j = (BatchJob *)arg;
  const size_t stride = (size_t)ORC_PT * j->n;
  for (;;) {
    const int b = atomic_fetch_add_explicit(j->next, 1, memory_order_relaxed);
    if (b >= j->B) break;
    double c;
    int st, it, nb, nf;
    orc_solve(j->s, j->init + b * stride, j->n, j->out_traj + b * stride, &c, &st, &it, &nb, &nf,
              0, 0, 0, 0);
    if (j->out_cost) j->out_cost[b] = c;
    if (j->out_status) j->out_status[b] = st;
    if (j->out_iters) j->out_iters[b] = it;
    if (j->out_n_bwd) j->out_n_bwd[b] = nb;
    if (j->out_n_fwd) j->out_n_fwd[b] = nf;
  }
  return 0;
}

int orc_solve_batch(const orc_solver *s, const double *init, int B, int n, double *out_traj,
                    double *out_cost, int *out_status, int *out_iters, int *out_n_bwd,
                    int *out_n_fwd, int n_threads) {
  if (n <= 0 || B < 0) return ORC_ERR_INVALID;
  if (n > s->n_desired) return ORC_ERR_LENGTH_MISMATCH;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 512) n_threads = 512;
  pthread_t th[512];
  atomic_int next;
  atomic_init(&next, 0);
  BatchJob job = {s, init, B, n, &next, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd};
  if (n_threads == 1) {
    batch_worker(&job);
    return ORC_OK;
  }
  int started = 0;
  for (int t = 0; t < n_threads; ++t)
    if (pthread_create(&th[started], 0, batch_worker, &job) == 0) ++started;
  if (started == 0) batch_worker(&job); /* no thread could be created: this one does the work */
  for (int t = 0; t < started; ++t) pthread_join(th[t], 0);
  return ORC_OK;
}

/* how this library was compiled: "parity" (-ffp-contract=off, the build every test compares with) or "fast" (timing only:
 * -O3 -mfma -ffp-contract=fast; bench.py's cpu_baseline) */
const char *orc_build_flavour(void) {
#ifdef ORC_FAST_BUILD
  return "fast: " ORC_FAST_BUILD;
#else
  return "parity: -O3 -march=x86-64-v3 -ffp-contract=off";
#endif
}
