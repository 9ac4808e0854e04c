// se3_math.h -- scalar SO(3)/SE(3) arithmetic and the per-knot quadrotor functions.
//
// Every function here is lane-local (no cross-lane traffic, no LDS) so that the same code
// serves a thread that owns one knot (k_linearize) and a thread that owns one problem
// (k_rollout).  Written from the closed forms of Sola et al., "A micro Lie theory" and
// Barfoot's SE(3) Q block with the branch structure manif uses (theta^2 <= 1e-10
// small-angle switches, atan2 log with the w<0 branch), because the reference calls manif
// at quadrotor_model.cc:183-186, 204, 211, 217, 232-235.  Where a product can be formed
// without materialising zero blocks it is; results differ from the dense reference
// formulation at rounding level only.
//
// QILQR_HD expands to __host__ __device__ under hipcc so that tests/host_harness.cpp can
// compile these functions with g++ and check them against the oracle on the CPU box;
// the product library never runs them on the host.
#pragma once

#if defined(__HIPCC__)
#define QILQR_HD __host__ __device__ __forceinline__
#else
#define QILQR_HD inline
#endif

#include <math.h>

// Keeps the instruction scheduler from hoisting every scalar load of the weight matrices to the top of
// a function (more constants than scalar registers: they would be spilled lane by lane).
// QILQR_REFETCH(): what was read from memory before this point is read again after it rather than kept in
// registers (the weights are used in several passes; keeping all of them live spills).
// QILQR_PIN(x): x is computed here, not sunk to its first use (which would leave its operands live).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QILQR_NO_SCHED_FENCES)
#define QILQR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define QILQR_REFETCH() asm volatile("" ::: "memory")
#define QILQR_PIN(x) asm volatile("" : "+v"(x))
#else
#define QILQR_SCHED_FENCE() do { } while (0)
#define QILQR_REFETCH() do { } while (0)
#define QILQR_PIN(x) do { } while (0)
#endif

// Fused multiply-adds are formed where the SOURCE says a * b + c in one expression and nowhere else (hipcc's default, "fast", also
// fuses across statements, and what it finds depends on the kernel a function is inlined into: the same linearisation inlined into two
// kernels gave records that differed in their last bits -- DESIGN.md section 4, k_round).  With this every function of this file
// computes the same bits in every kernel that uses it; the rest of the translation unit goes back to the default below.
#if defined(__clang__)
#pragma clang fp contract(on)
#endif

namespace qilqr {

template <typename T>
struct Eps {
  static constexpr T manif = T(1e-10);  // manif Constants<double>::eps
};
template <>
struct Eps<float> {
  static constexpr float manif = 1e-6f;  // manif Constants<float>::eps (+: recollection of upstream)
};

// ----------------------------------------------------------------- batch layouts
// Per-knot arrays (trajectories W = 18, gains W = 52, knot costs W = 1, knot records) are stored "tiled": trajectories are
// grouped TILE at a time and, inside a tile, the trajectory index is the fastest dimension in units of 16 bytes:
//     [tile = b / TILE][knot i][pair e / 2][slot = b % TILE][e % 2]
// TILE = 4 is the ownership of the kernels that carry a solve: k_backward4, k_rollout16 and k_solve4 give a workgroup to
// four consecutive trajectories, so a knot of a tile (gains: 1664 bytes, records: 2944, trajectories: 576) is one
// contiguous piece of memory that exactly one workgroup reads or writes, whole cache lines at a time.  The
// lane-per-trajectory kernels (k_linearize, k_rollout, k_rollout3, the format conversions) see 64-byte pieces, sixteen per
// instruction.
// (TILE = 64, the first design, gave those kernels one contiguous kilobyte per instruction, but the sixteen workgroups
// of a tile then shared every line, 32 bytes each, and so did the 64 one-wavefront workgroups of the largest batches.
// Same box, whole solves, TILE 64 -> 4: B = 1024 +1 %, 8192 +4 %, 65536 +11 %; k_backward4 fetches 38.8 MB per launch
// instead of 44.4 and k_rollout16 26.6 instead of 30.5, while k_linearize takes 189 us instead of 166 at B = 8192.
// TILE = 8 measures like 64.  Build with -DQILQR_TILE=64 to compare: make variant NAME=t64 DEFS=-DQILQR_TILE=64.)
// TILED = false is the plain [b][i][W] layout of the C ABI (and of the CPU harness).
#ifndef QILQR_TILE
#define QILQR_TILE 4
#endif
constexpr int TILE = QILQR_TILE, TILE2 = 2 * TILE;  // (TILE2: elements between consecutive entry pairs of a trajectory)
constexpr int TILE_LOG = TILE == 64 ? 6 : TILE == 32 ? 5 : TILE == 16 ? 4 : TILE == 8 ? 3 : TILE == 4 ? 2 : -1;
static_assert(TILE_LOG > 0 && (1 << TILE_LOG) == TILE, "TILE is 4, 8, 16, 32 or 64 (sub-batches start at multiples of 64)");
template <bool TILED>
QILQR_HD long knot_base(long b, long n, int W) {
  return TILED ? (b >> TILE_LOG) * n * (W >> 1) * TILE2 + ((b & (TILE - 1)) << 1) : b * n * W;
}
template <bool TILED>
QILQR_HD long knot_elem(long i, int e, int W) {
  return TILED ? (i * (W >> 1) + (e >> 1)) * TILE2 + (e & 1) : i * W + e;
}
template <bool TILED, typename T>
QILQR_HD void load_knot(const T *base, long i, int W, T *dst) {
#pragma unroll
  for (int e = 0; e < 52; ++e)
    if (e < W) dst[e] = base[knot_elem<TILED>(i, e, W)];
}
QILQR_HD long tiled_count(long B, long n, int W) { return ((B + 63) >> 6) * 64 * n * W; }
// knot costs: [tile][i][slot]
QILQR_HD long cost_index(long b, long i, long n) { return ((b >> TILE_LOG) * n + i) * TILE + (b & (TILE - 1)); }

// ----------------------------------------------------------------- 3-vectors / 3x3 (row-major)
template <typename T>
QILQR_HD void skew3(const T a[3], T S[9]) {
  S[0] = T(0); S[1] = -a[2]; S[2] = a[1];
  S[3] = a[2]; S[4] = T(0);  S[5] = -a[0];
  S[6] = -a[1]; S[7] = a[0]; S[8] = T(0);
}
template <typename T>
QILQR_HD void mat3_mul(const T A[9], const T B[9], T C[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <typename T>
QILQR_HD void mat3_vec(const T A[9], const T v[3], T o[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
template <typename T>
QILQR_HD void mat3_tvec(const T A[9], const T v[3], T o[3]) {  // A^T v
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
template <typename T>
QILQR_HD void cross3(const T a[3], const T b[3], T o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// ----------------------------------------------------------------- quaternions (x,y,z,w)
// rotation matrix of a unit quaternion (Eigen's toRotationMatrix form)
template <typename T>
QILQR_HD void quat_to_R(const T q[4], T R[9]) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const T twx = tx * w, twy = ty * w, twz = tz * w;
  const T txx = tx * x, txy = ty * x, txz = tz * x;
  const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
template <typename T>
QILQR_HD void quat_mul(const T a[4], const T b[4], T o[4]) {
#if defined(__clang__)
#pragma clang fp contract(off)  // conj(q) * q must give an exactly zero vector part (x (-) x == 0)
#endif
  const T ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const T bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

// the same product where exactness of conj(q) * q does not matter (pose composition): the compiler
// may fuse the multiply-adds (16 instructions instead of 28 on the rollout's serial chain)
template <typename T>
QILQR_HD void quat_mul_fused(const T a[4], const T b[4], T o[4]) {
  const T ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const T bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

// ----------------------------------------------------------------- SO(3)
// Exp: [sin(|th|/2) th/|th| ; cos(|th|/2)], small angle [th/2 ; 1]
template <typename T>
QILQR_HD void so3_exp(const T th[3], T q[4]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  if (th2 > Eps<T>::manif) {
    const T theta = sqrt(th2);
    const T ha = T(0.5) * theta;
    const T s = sin(ha), c = cos(ha);
    q[0] = s * (th[0] / theta); q[1] = s * (th[1] / theta); q[2] = s * (th[2] / theta); q[3] = c;
  } else {
    q[0] = th[0] / 2; q[1] = th[1] / 2; q[2] = th[2] / 2; q[3] = T(1);
  }
}
// Log: 2 atan2(|q_v|, w) q_v/|q_v| with the w<0 branch, small angle 2 q_v
template <typename T>
QILQR_HD void so3_log(const T q[4], T th[3]) {
  const T s2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  T coeff;
  if (s2 > Eps<T>::manif) {
    const T s = sqrt(s2);
    const T c = q[3];
    const T two_angle = T(2) * ((c < T(0)) ? atan2(-s, -c) : atan2(s, c));
    coeff = two_angle / s;
  } else {
    coeff = T(2);
  }
  th[0] = q[0] * coeff; th[1] = q[1] * coeff; th[2] = q[2] * coeff;
}
// coefficients (a, b) of  Jl = I + a W + b W^2   (Jr = Jl^T = I - a W + b W^2)
template <typename T>
QILQR_HD void so3_jac_coeffs(T th2, T &a, T &b, bool &small) {
  small = !(th2 > Eps<T>::manif);
  if (small) {
    a = T(0.5);
    b = T(0);
  } else {
    const T theta = sqrt(th2);
    a = (T(1) - cos(theta)) / th2;
    b = (theta - sin(theta)) / (th2 * theta);
  }
}
// Jl(th) as a matrix
template <typename T>
QILQR_HD void so3_ljac(const T th[3], T J[9]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T a, b;
  bool small;
  so3_jac_coeffs(th2, a, b, small);
  T W[9], WW[9];
  skew3(th, W);
  mat3_mul(W, W, WW);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = a * W[i] + (small ? T(0) : b * WW[i]);
  J[0] += 1; J[4] += 1; J[8] += 1;
}
// Jl^-1(th) = I - W/2 + c W^2,  c = 1/th^2 - (1+cos)/(2 th sin)
template <typename T>
QILQR_HD void so3_ljacinv(const T th[3], T J[9]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T W[9], WW[9];
  skew3(th, W);
  if (!(th2 > Eps<T>::manif)) {
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = T(-0.5) * W[i];
  } else {
    const T theta = sqrt(th2);
    const T c = T(1) / th2 - (T(1) + cos(theta)) / (T(2) * theta * sin(theta));
    mat3_mul(W, W, WW);
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = T(-0.5) * W[i] + c * WW[i];
  }
  J[0] += 1; J[4] += 1; J[8] += 1;
}
template <typename T>
QILQR_HD void transpose3(const T A[9], T At[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) At[3 * i + j] = A[3 * j + i];
}

// ----------------------------------------------------------------- SE(3), pose = (t[3], q[4])
// Barfoot's Q block at c = [rho ; theta]
template <typename T>
QILQR_HD void se3_fillQ(const T rho[3], const T th[3], T Qm[9]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T B, C, D;
  if (!(th2 > Eps<T>::manif)) {
    B = T(1. / 6.) - th2 / T(120);
    C = T(-1. / 24.) + th2 / T(720);
    D = T(-1. / 120.);
  } else {
    const T theta = sqrt(th2);
    const T s = sin(theta), co = cos(theta);
    B = (theta - s) / (th2 * theta);
    C = (T(1) - th2 / T(2) - co) / (th2 * th2);
    D = T(0.5) * (C - T(3) * (theta - s - th2 * theta / T(6)) / (th2 * th2 * theta));
  }
  T V[9], W[9], VW[9], WV[9], WVW[9], VWW[9], WW[9], WWV[9], WVWW[9], WWVW[9];
  skew3(rho, V);
  skew3(th, W);
  mat3_mul(V, W, VW);
  mat3_mul(W, V, WV);
  mat3_mul(WV, W, WVW);
  mat3_mul(VW, W, VWW);
  mat3_mul(W, W, WW);
  mat3_mul(WW, V, WWV);
  mat3_mul(WVW, W, WVWW);
  mat3_mul(WW, VW, WWVW);
#pragma unroll
  for (int i = 0; i < 9; ++i)
    Qm[i] = T(0.5) * V[i] + B * (WV[i] + VW[i] + WVW[i]) - C * (WWV[i] + VWW[i] - T(3) * WVW[i]) -
            D * (WVWW[i] + WWVW[i]);
}

// T_out = T * Exp(tau), tau = [rho ; th]
template <typename T>
QILQR_HD void se3_rplus(const T t[3], const T q[4], const T tau[6], T to[3], T qo[4]) {
  T Jl[9], p[3], qe[4], R[9], Rp[3];
  so3_ljac(tau + 3, Jl);
  mat3_vec(Jl, tau, p);
  so3_exp(tau + 3, qe);
  quat_to_R(q, R);
  mat3_vec(R, p, Rp);
  quat_mul(q, qe, qo);
  const T n = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
  if (fabs(n - T(1)) > Eps<T>::manif) {  // manif's compose renormalisation
    const T sc = T(2) / (T(1) + n);
    qo[0] *= sc; qo[1] *= sc; qo[2] *= sc; qo[3] *= sc;
  }
  to[0] = Rp[0] + t[0]; to[1] = Rp[1] + t[1]; to[2] = Rp[2] + t[2];
}

// tau = Log(X^-1 Y)   (Y (-) X)
template <typename T>
QILQR_HD void se3_rminus(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6]) {
  // X^-1 = (-R(qx*) tx, qx*)
  const T qc[4] = {-qx[0], -qx[1], -qx[2], qx[3]};
  T Rc[9], a[3], b[3], qd[4];
  quat_to_R(qc, Rc);
  mat3_vec(Rc, tx, a);
  mat3_vec(Rc, ty, b);
  T td[3] = {b[0] + (-a[0]), b[1] + (-a[1]), b[2] + (-a[2])};
  quat_mul(qc, qy, qd);
  const T n = qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3];
  if (fabs(n - T(1)) > Eps<T>::manif) {
    const T sc = T(2) / (T(1) + n);
    qd[0] *= sc; qd[1] *= sc; qd[2] *= sc; qd[3] *= sc;
  }
  T th[3], Ji[9];
  so3_log(qd, th);
  so3_ljacinv(th, Ji);
  mat3_vec(Ji, td, tau);
  tau[3] = th[0]; tau[4] = th[1]; tau[5] = th[2];
}

// ----------------------------------------------------------------- model constants on the device
// Built on the host in qilqr_create (quadrotor_model.cc:6-25 + the constant parts of
// continuous_dynamics :113-119); passed to every kernel by value.
template <typename T>
struct ModelConsts {
  T dt, mass, g;
  T inertia[9];
  T inertia_inv[9];
  T arms[12];  // moment_arms 3x4
  T Bu[48];    // J_u (12x4), constant: row 8 = dt/m, rows 9-11 = dt I^-1 moment_arms
  T Q[144];
  T R[16];
};

// the same constants in another precision (mixed-precision mode: lane-local kernels run in float)
template <typename T, typename U>
QILQR_HD void convert_consts(const ModelConsts<U> &a, ModelConsts<T> &b) {
  b.dt = (T)a.dt; b.mass = (T)a.mass; b.g = (T)a.g;
  for (int i = 0; i < 9; ++i) { b.inertia[i] = (T)a.inertia[i]; b.inertia_inv[i] = (T)a.inertia_inv[i]; }
  for (int i = 0; i < 12; ++i) b.arms[i] = (T)a.arms[i];
  for (int i = 0; i < 48; ++i) b.Bu[i] = (T)a.Bu[i];
  for (int i = 0; i < 144; ++i) b.Q[i] = (T)a.Q[i];
  for (int i = 0; i < 16; ++i) b.R[i] = (T)a.R[i];
}

// continuous dynamics (quadrotor_model.cc:65-78) -> body acceleration (6)
template <typename T>
QILQR_HD void body_acceleration(const ModelConsts<T> &c, const T q[4], const T v[6], const T u[4],
                                T acc[6]) {
  T R[9];
  quat_to_R(q, R);
  const T usum = ((u[0] + u[1]) + u[2]) + u[3];
  acc[0] = -c.g * R[6];
  acc[1] = -c.g * R[7];
  acc[2] = -c.g * R[8] + usum / c.mass;  // no -omega x v term (quadrotor_model.cc:69-72)
  T M[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    M[i] = c.arms[4 * i] * u[0] + c.arms[4 * i + 1] * u[1] + c.arms[4 * i + 2] * u[2] +
           c.arms[4 * i + 3] * u[3];
  const T *w = v + 3;
  T Iw[3], wIw[3], rhs[3];
  mat3_vec(c.inertia, w, Iw);
  cross3(w, Iw, wIw);
#pragma unroll
  for (int i = 0; i < 3; ++i) rhs[i] = M[i] - wIw[i];
  mat3_vec(c.inertia_inv, rhs, acc + 3);
}

// one explicit-Euler step on SE(3) x R^6 (quadrotor_model.cc:33-49, 266-276)
template <typename T>
QILQR_HD void discrete_step(const ModelConsts<T> &c, T t[3], T q[4], T v[6], const T u[4]) {
  T acc[6], tau[6], tn[3], qn[4];
  body_acceleration(c, q, v, u, acc);
#pragma unroll
  for (int i = 0; i < 6; ++i) tau[i] = c.dt * v[i];  // pose integrates with the OLD velocity
  se3_rplus(t, q, tau, tn, qn);
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = tn[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = qn[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = v[i] + c.dt * acc[i];
}

template <typename T>
QILQR_HD void se3_rminus_fast(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6]);
// What the cost half of the linearisation needs of a rotation error theta three times over -- for J_l^-1 inside x (-) x_d,
// for the blocks of J_r^-1 and for Barfoot's Q block: formed ONCE, where the Log is taken (se3_rminus_part1).  Beyond the
// series' range (half a radian: every early knot of a random start) each of the three used to take its own square root,
// sine, cosine and divisions of the same angle: 8.4 us for a lone wavefront on that path against 4.1 on the series'.
template <typename T>
struct RotScalars {
  T th2, c;          // theta^2 as the Log formed it; the coefficient of J^-1 = I -+ W/2 + c W^2
  T theta, sn, cs;   // theta, sin theta, cos theta: defined when trig is set
  bool trig;         // theta^2 beyond the range of the series (EXP_MAX)
};
template <typename T>
QILQR_HD void se3_rminus_fast(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6], RotScalars<T> &rs);
template <typename T>
struct Series;
template <typename T>
QILQR_HD T poly8(const T c[8], T x);

// ----------------------------------------------------------------- matrices for the linearisation
// The same Jacobians as so3_ljac / so3_ljacinv / se3_fillQ, formed without 3x3 matrix products from
//   hat(a) hat(b) = b a^T - (a.b) I ,   W^2 = th th^T - |th|^2 I
// and with the series of the scalar coefficients for moderate angles (manif's theta^2 <= 1e-10
// branches kept).  k_linearize is bound by its instruction count; this is a third of it.
template <typename T>
QILQR_HD void so3_sym_part(const T th[3], T th2, T coef, T J[9]) {  // J += coef (th th^T - th2 I)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) J[3 * i + j] += coef * (th[i] * th[j] - ((i == j) ? th2 : T(0)));
}
template <typename T>
QILQR_HD void so3_ljac_fast(const T th[3], T J[9]) {  // I + a W + b W^2
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T a, b;
  if (!(th2 > Eps<T>::manif)) {
    a = T(0.5); b = T(0);
  } else if (th2 <= Series<T>::EXP_MAX) {
    a = poly8(Series<T>::jac_a, th2);
    b = poly8(Series<T>::jac_b, th2);
  } else {
    const T theta = sqrt(th2);
    a = (T(1) - cos(theta)) / th2;
    b = (theta - sin(theta)) / (th2 * theta);
  }
  skew3(th, J);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] *= a;
  J[0] += 1; J[4] += 1; J[8] += 1;
  so3_sym_part(th, th2, b, J);
}
template <typename T>
QILQR_HD void so3_ljacinv_fast(const T th[3], T J[9]) {  // I - W/2 + c W^2
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T c;
  if (!(th2 > Eps<T>::manif)) {
    c = T(0);
  } else if (th2 <= Series<T>::JINV_MAX) {
    c = poly8(Series<T>::jinv_c, th2);
  } else {
    const T theta = sqrt(th2);
    c = T(1) / th2 - (T(1) + cos(theta)) / (T(2) * theta * sin(theta));
  }
  skew3(th, J);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] *= T(-0.5);
  J[0] += 1; J[4] += 1; J[8] += 1;
  so3_sym_part(th, th2, c, J);
}
// I - W/2 + c W^2 with the coefficient c(theta^2) already known (it is an even function of theta: the same for th and -th)
template <typename T>
QILQR_HD void so3_ljacinv_with(const T th[3], T c, T J[9]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  skew3(th, J);
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] *= T(-0.5);
  J[0] += 1; J[4] += 1; J[8] += 1;
  so3_sym_part(th, th2, c, J);
}
template <typename T>
QILQR_HD void se3_fillQ_body(const T rho[3], const T th[3], T th2, T B, T C, T D, T Qm[9]);
// se3_fillQ_fast with theta, sin theta and cos theta taken from where the Log was formed (rs belongs to th or to -th)
template <typename T>
QILQR_HD void se3_fillQ_with(const T rho[3], const T th[3], const RotScalars<T> &rs, T Qm[9]) {
  const T th2 = rs.th2;
  T B, C, D;
  if (!(th2 > Eps<T>::manif)) {
    B = T(1. / 6.) - th2 / T(120);
    C = T(-1. / 24.) + th2 / T(720);
    D = T(-1. / 120.);
  } else if (!rs.trig) {
    B = poly8(Series<T>::jac_b, th2);
    C = poly8(Series<T>::fillq_C, th2);
    D = poly8(Series<T>::fillq_D, th2);
  } else {
    const T theta = rs.theta, s = rs.sn, co = rs.cs;
    B = (theta - s) / (th2 * theta);
    C = (T(1) - th2 / T(2) - co) / (th2 * th2);
    D = T(0.5) * (C - T(3) * (theta - s - th2 * theta / T(6)) / (th2 * th2 * theta));
  }
  se3_fillQ_body(rho, th, th2, B, C, D, Qm);
}
// Q(rho, th) = V/2 + B (rho th^T + th rho^T) - 2 B s I - (B + C) s W - C (n th^T - th n^T) + 2 D s W^2,
// s = th.rho, n = th x rho  (Barfoot eq. 102 with the products of skew matrices written out)
template <typename T>
QILQR_HD void se3_fillQ_fast(const T rho[3], const T th[3], T Qm[9]) {
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T B, C, D;
  if (!(th2 > Eps<T>::manif)) {
    B = T(1. / 6.) - th2 / T(120);
    C = T(-1. / 24.) + th2 / T(720);
    D = T(-1. / 120.);
  } else if (th2 <= Series<T>::EXP_MAX) {
    B = poly8(Series<T>::jac_b, th2);
    C = poly8(Series<T>::fillq_C, th2);
    D = poly8(Series<T>::fillq_D, th2);
  } else {
    const T theta = sqrt(th2);
    const T s = sin(theta), co = cos(theta);
    B = (theta - s) / (th2 * theta);
    C = (T(1) - th2 / T(2) - co) / (th2 * th2);
    D = T(0.5) * (C - T(3) * (theta - s - th2 * theta / T(6)) / (th2 * th2 * theta));
  }
  se3_fillQ_body(rho, th, th2, B, C, D, Qm);
}
template <typename T>
QILQR_HD void se3_fillQ_body(const T rho[3], const T th[3], T th2, T B, T C, T D, T Qm[9]) {
  const T s = th[0] * rho[0] + th[1] * rho[1] + th[2] * rho[2];
  T n[3], V[9], W[9];
  cross3(th, rho, n);
  skew3(rho, V);
  skew3(th, W);
  const T cw = -(B + C) * s, cd = T(2) * D * s, ci = T(-2) * B * s;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      T q = T(0.5) * V[3 * i + j] + B * (rho[i] * th[j] + th[i] * rho[j]) + cw * W[3 * i + j] -
            C * (n[i] * th[j] - th[i] * n[j]) + cd * (th[i] * th[j]);
      if (i == j) q += ci - cd * th2;
      Qm[3 * i + j] = q;
    }
}

// ----------------------------------------------------------------- knot record layout
// What k_linearize hands to k_backward for one knot (doubles).  Only entries that are not
// structurally constant are stored; the record is the dominant HBM traffic of the solver, so its
// size is chosen per problem class (RecLayout, built once on the host):
//   [  0.. 53] six 3x3 blocks of J_x, row-major:
//              0: E^T            (rows 0-2 x cols 0-2 and rows 3-5 x cols 3-5)
//              1: hat(-E^T p)E^T (rows 0-2 x cols 3-5)            Ad(Exp(tau)^-1)
//              2: dt Jr(theta)   (rows 0-2 x cols 6-8 and rows 3-5 x cols 9-11)
//              3: dt Q(-tau)     (rows 0-2 x cols 9-11)
//              4: -dt g hat(R^T e_z)                (rows 6-8 x cols 3-5)
//              5: I - dt I^-1 (hat(w) I - hat(I w)) (rows 9-11 x cols 9-11)
//   [off_cxx ..] C_xx = 2 J^T Q J (cost.hh:52), J = blkdiag(Jri, I6):
//              general Q   : all 144 entries, row-major
//              symmetric Q : rows 0..5, each from its diagonal entry to column 5 (upper triangle of the
//                            pose block, 21) plus columns 6..11 (pose x velocity block, 36) unless
//                            Q[0:6,6:12] == 0 (then that block is zero and not stored);
//                            the lower-right block is the constant 2 Q[6:12,6:12]
//   [off_g ..+15] C_x (12), C_u (4)         cost.hh:51,54
//   [off_cost]    knot cost                 cost.hh:47-48
constexpr int LIN_BLK = 0;
constexpr int LIN_MAX_STRIDE = 354;  // dense M (192) + general C_xx (144) + g (16) + cost, even

struct RecLayout {
  int sym;      // Q == Q^T exactly
  int ur_zero;  // sym and Q[0:6, 6:12] == 0
  int off_cxx, off_g, off_cost, stride;
  int dense_m;  // the record starts with the whole M = [J_x | J_u] (12 x 16, row-major) instead of the six Jacobian
                // blocks of the explicit-Euler step: the Runge-Kutta extension, whose Jacobians have no such structure
  int tiled;    // placement of the records in memory (not of the entries in a record): see rec_base / rec_elem.  Set per
                // call by the host to what the call's backward kernel reads
};
constexpr int LIN_M_DENSE = 192, LIN_M_BLOCKS = 54;
QILQR_HD constexpr RecLayout make_layout(bool sym, bool ur_zero, bool dense_m = false) {
  RecLayout L{};
  L.sym = sym ? 1 : 0;
  L.ur_zero = (sym && ur_zero) ? 1 : 0;
  L.dense_m = dense_m ? 1 : 0;
  L.off_cxx = dense_m ? LIN_M_DENSE : LIN_M_BLOCKS;
  const int ncxx = !sym ? 144 : (L.ur_zero ? 21 : 57);
  L.off_g = L.off_cxx + ncxx;
  L.off_cost = L.off_g + 16;
  L.stride = (L.off_cost + 1 + 1) & ~1;  // even: records stay 16-byte aligned
  return L;
}
// Device layout of the knot records, two forms.
// PLAIN  [b][knot][stride]: one contiguous record per knot.  What the one-wavefront backward kernel reads (seven entries per
//   lane and knot through per-lane pointers: a trajectory's knot is six cache lines).
// TILED  [tile of TILE trajectories][knot][entry pair][slot][2]: the layout of the trajectories and gains.  What k_linearize
//   writes best -- its 64 lanes hold one knot of 64 consecutive trajectories, so a 16-byte store per lane fills sixteen
//   64-byte pieces instead of touching 64 different lines (the kernel was bound by the L2's request rate, not by bytes:
//   36 -> 24 us per launch at B = 1024 with every trajectory live, 325 -> 189 at B = 8192) -- and what the kernels that
//   stage records through LDS read (k_backward4, k_backward2, k_solve4: one 16-byte load per lane and record; a knot of
//   k_backward4's four trajectories is 23 consecutive lines).
// (Measured and dropped earlier: contiguous records written out through an LDS transpose -- padding, slower -- and the 64
// records of a wavefront next to each other, [tile][knot][lane][stride], which changes nothing: each lane's stores still
// go to a line of their own.)
QILQR_HD long rec_base(const RecLayout &L, long b, long n) {
  return L.tiled ? (b >> TILE_LOG) * n * L.stride * TILE + (b & (TILE - 1)) * 2 : b * n * L.stride;
}
QILQR_HD long rec_elem(const RecLayout &L, long i, int k) {
  return L.tiled ? (i * (L.stride / 2) + (k >> 1)) * TILE2 + (k & 1) : i * L.stride + k;
}
QILQR_HD long rec_count(long B, long n, int stride) { return ((B + 63) / 64) * 64 * n * stride; }  // (room for either form)
// The linearisation hands its entries to a writer: put(k, v) stores entry k of the record; flush() ends the record.
template <typename T>
struct PlainRecWriter {
  T *rec;
  QILQR_HD void put(int k, T v) const { rec[k] = v; }  // (non-temporal stores here are 6x slower: the L2 must merge them)
  QILQR_HD void flush() const {}
};
// the same for the tiled placement: rec points at entry 0 of the knot; entry pairs are TILE2 elements apart.  The entries
// arrive almost always in ascending order (the loops that produce them are unrolled, k is a constant at every call): an even
// entry waits in a register for its odd neighbour and the two leave as one 16-byte store -- half the store instructions,
// and each writes whole 64-byte pieces of the tiles' lines instead of every other eight bytes of them.
template <typename T>
struct TiledRecWriter {
  T *rec;
  mutable T pend = T(0);
  mutable int pend_k = -1;  // the even entry that is waiting, or -1
  QILQR_HD void put(int k, T v) const {
    if ((k & 1) == 0) {
      flush();
      pend = v;
      pend_k = k;
    } else if (pend_k == k - 1) {
#ifdef __HIPCC__
      typedef T pair_t __attribute__((ext_vector_type(2)));
      const pair_t pr = {pend, v};
      *reinterpret_cast<pair_t *>(rec + (k >> 1) * TILE2) = pr;
#else
      rec[(k >> 1) * TILE2] = pend;
      rec[(k >> 1) * TILE2 + 1] = v;
#endif
      pend_k = -1;
    } else {
      rec[(k >> 1) * TILE2 + 1] = v;
    }
  }
  QILQR_HD void flush() const {
    if (pend_k >= 0) rec[(pend_k >> 1) * TILE2] = pend;
    pend_k = -1;
  }
};
// symmetric layouts, rows i < 6 of C_xx in order of production: row i holds its part of the upper triangle
// of the pose block (columns i..5) and, when the pose x velocity block is stored, its six entries of that
QILQR_HD constexpr int symrow_index(bool with_pv, int i, int j) {  // i < 6, i <= j < (with_pv ? 12 : 6)
  const int w = with_pv ? 12 : 6;
  return i * w - (i * (i - 1)) / 2 + (j - i);
}
QILQR_HD constexpr int sym6_index(int i, int j) {  // i <= j < 6, packed upper triangle, row-major
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}
// where C_xx[row][col] (row, col < 12) lives in a record: offset, or -1 with *cst = the constant
QILQR_HD int cxx_source(const RecLayout &L, int row, int col, const double *Q, double *cst) {
  *cst = 0.0;
  if (!L.sym) return L.off_cxx + row * 12 + col;
  const int i = row < col ? row : col, j = row < col ? col : row;
  if (j < 6) return L.off_cxx + symrow_index(!L.ur_zero, i, j);
  if (i < 6) return L.ur_zero ? -1 : L.off_cxx + symrow_index(true, i, j);
  *cst = 2.0 * Q[row * 12 + col];
  return -1;
}

// knot value of the cost (cost.hh:36-48); pt/pd = 18-double knots.  Also returns the row vectors
// sq = dx^T Q and sr = du^T R, which the differentials reuse (C_x = 2 sq J, C_u = 2 sr).
// Q (12x12) and R (4x4) are read row by row through pointers (k_linearize keeps them in LDS).
// BLOCKDIAG: Q's pose x velocity blocks are exactly zero, their products are skipped (adding exact
// zeros changes nothing for finite dx).
// DIAG: Q is exactly diagonal (the reference's demo and tests: Q = diag): sq[j] = dx[j] Q[j][j].  The general sum would add
// products with exact zeros to that one product, which changes nothing: the same bits (up to the sign of a zero).
template <bool BLOCKDIAG, typename T, bool DIAG = false>
QILQR_HD T knot_cost(const T *Q, const T *R, const T *pt, const T *pd, T dx[12], T du[4], T sq[12], T sr[4],
                     RotScalars<T> *rs = nullptr) {
  const T qx[4] = {pt[5], pt[6], pt[7], pt[4]};
  const T qd[4] = {pd[5], pd[6], pd[7], pd[4]};
  // x (-) x_d; exactly zero at zero error
  if (rs) se3_rminus_fast(pt + 1, qx, pd + 1, qd, dx, *rs);
  else se3_rminus_fast(pt + 1, qx, pd + 1, qd, dx);
#pragma unroll
  for (int i = 0; i < 6; ++i) dx[6 + i] = pt[8 + i] - pd[8 + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) du[i] = pt[14 + i] - pd[14 + i];
  // sq[j] = sum_i dx[i] Q[i][j], i ascending for every j
  if constexpr (DIAG) {
#pragma unroll
    for (int j = 0; j < 12; ++j) sq[j] = dx[j] * Q[j * 12 + j];
  } else {
#pragma unroll
    for (int j = 0; j < 12; ++j) sq[j] = T(0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int j0 = BLOCKDIAG ? (i < 6 ? 0 : 6) : 0, j1 = BLOCKDIAG ? j0 + 6 : 12;
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if (j >= j0 && j < j1) sq[j] += dx[i] * Q[i * 12 + j];
      if ((i & 1) == 1) QILQR_SCHED_FENCE();  // two rows of weights in flight, not all twelve
    }
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) QILQR_PIN(sq[j]);
  T cx = T(0);
#pragma unroll
  for (int j = 0; j < 12; ++j) cx += sq[j] * dx[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) sr[j] = T(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sr[j] += du[i] * R[i * 4 + j];
  T cu = T(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) cu += sr[j] * du[j];
  return cx + cu;
}

// ----------------------------------------------------------------- the Runge-Kutta extension
// The step sketched in the comment at quadrotor_model.cc:51-63 (never executed by the reference; default off here):
//     k_0 = f(x, u);  k_i = f(x (+) h_i k_{i-1}, u), h = {0, dt/2, dt/2, dt};  x_next = x (+) dt (k_0 + 2 k_1 + 2 k_2 + k_3) / 6
// with (+) = euler_step (quadrotor_model.cc:266-276) from x in every stage.  The oracle states it from the reference's
// primitives and measures its order on SE(3): two (tests/test_oracle_rk4.py).
// Jacobians by the chain rule, with the sparsity of the primitives written out: for a tangent step tau = h k[0:6],
//     d(x (+) h k)/dx = [[Ad(Exp(-tau)), 0],[0, 1]],   d(x (+) h k)/dk = h [[Jr(tau), 0],[0, 1]],
//     F_x = df/dx = [[0, 1],[G, D]] with G = -g hat(R^T e_z) in the rotation columns of the linear rows (quadrotor_model.cc:88-96)
//     and D = -I^-1 (hat(w) I - hat(I w)) in the angular block (:99-111),   F_u = J_u / dt (rows 8..11).
// MU (12 x 16) carries [d./dx | d./du] through the stages.
template <typename T>
struct ExpBlocks {  // for tau: Ad(Exp(-tau)) = [[Rc, SR],[0, Rc]],  Jr(tau) = [[Jr, Qm],[0, Jr]]
  T Rc[9], SR[9], Jr[9], Qm[9];
};
template <typename T>
QILQR_HD void exp_blocks(const T tau[6], ExpBlocks<T> &e) {
  T Jl[9], p[3], qe[4];
  so3_ljac_fast(tau + 3, Jl);
  mat3_vec(Jl, tau, p);
  so3_exp(tau + 3, qe);
  const T qc[4] = {-qe[0], -qe[1], -qe[2], qe[3]};
  T r[3], ti[3], S[9];
  quat_to_R(qc, e.Rc);
  mat3_vec(e.Rc, p, r);
  ti[0] = -r[0]; ti[1] = -r[1]; ti[2] = -r[2];
  skew3(ti, S);
  mat3_mul(S, e.Rc, e.SR);
  const T nrho[3] = {-tau[0], -tau[1], -tau[2]}, nth[3] = {-tau[3], -tau[4], -tau[5]};
  se3_fillQ_fast(nrho, nth, e.Qm);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) e.Jr[3 * i + j] = Jl[3 * j + i];
}
// G and D of F_x at (q, v)
template <typename T>
QILQR_HD void continuous_blocks(const ModelConsts<T> &c, const T q[4], const T v[6], T G[9], T D[9]) {
  T R[9];
  quat_to_R(q, R);
  const T rz[3] = {R[6], R[7], R[8]};
  T H[9];
  skew3(rz, H);
  for (int i = 0; i < 9; ++i) G[i] = -c.g * H[i];
  const T *om = v + 3;
  T Wh[9], WI[9], Iw[3], IwH[9], Jd[9], S[9];
  skew3(om, Wh);
  mat3_mul(Wh, c.inertia, WI);
  mat3_vec(c.inertia, om, Iw);
  skew3(Iw, IwH);
  for (int i = 0; i < 9; ++i) Jd[i] = WI[i] - IwH[i];
  mat3_mul(c.inertia_inv, Jd, S);
  for (int i = 0; i < 9; ++i) D[i] = -S[i];
}
// out (6 x 16) = [[A, B],[0, A]] in (6 x 6, blocks 3 x 3) times rows r0..r0+5 of X (12 x 16)
template <typename T>
QILQR_HD void blk6_mul(const T A[9], const T B[9], const T *X, T *out) {
  for (int i = 0; i < 3; ++i)
    for (int col = 0; col < 16; ++col) {
      T top = T(0), bot = T(0);
      for (int k = 0; k < 3; ++k) {
        top += A[3 * i + k] * X[k * 16 + col] + B[3 * i + k] * X[(3 + k) * 16 + col];
        bot += A[3 * i + k] * X[(3 + k) * 16 + col];
      }
      out[i * 16 + col] = top;
      out[(3 + i) * 16 + col] = bot;
    }
}
// One step from (t, q, v) under u; MU != nullptr: also M = [J_x | J_u] (12 x 16, row-major).
template <typename T>
QILQR_HD void rk4_step(const ModelConsts<T> &c, T t[3], T q[4], T v[6], const T u[4], T *MU) {
  const T coeffs[4] = {T(1.0 / 6.0), T(2.0 / 6.0), T(2.0 / 6.0), T(1.0 / 6.0)};
  const T hs[4] = {T(0), c.dt / T(2), c.dt / T(2), c.dt};
  T k[12], xdot[12];
  T K[192], SK[192];  // [dk/dx | dk/du] of the last stage, and its weighted sum
  for (int e = 0; e < 12; ++e) { k[e] = T(0); xdot[e] = T(0); }
  if (MU)
    for (int e = 0; e < 192; ++e) { K[e] = T(0); SK[e] = T(0); }
  for (int i = 0; i < 4; ++i) {
    // x_i = x (+) h k
    T tau[6], ti[3], qi[4], vi[6];
    for (int a = 0; a < 6; ++a) tau[a] = hs[i] * k[a];
    se3_rplus(t, q, tau, ti, qi);
    for (int a = 0; a < 6; ++a) vi[a] = v[a] + hs[i] * k[6 + a];
    T A[192];  // [dx_i/dx | dx_i/du]
    if (MU) {
      ExpBlocks<T> eb;
      exp_blocks(tau, eb);
      T JK[96];
      blk6_mul(eb.Jr, eb.Qm, K, JK);  // Jr(tau) (dk/d.)[pose rows]
      for (int r = 0; r < 6; ++r)
        for (int col = 0; col < 16; ++col) {
          T ad = T(0);  // Ad(Exp(-tau))[r][col], col < 6
          if (col < 6) {
            const int br = r / 3, bc = col / 3, rr = r % 3, cc = col % 3;
            ad = (br == bc) ? eb.Rc[3 * rr + cc] : (br == 0 ? eb.SR[3 * rr + cc] : T(0));
          }
          A[r * 16 + col] = ad + hs[i] * JK[r * 16 + col];
          A[(6 + r) * 16 + col] = ((col == 6 + r) ? T(1) : T(0)) + hs[i] * K[(6 + r) * 16 + col];
        }
    }
    // k = f(x_i, u)
    T acc[6];
    body_acceleration(c, qi, vi, u, acc);
    for (int a = 0; a < 6; ++a) { k[a] = vi[a]; k[6 + a] = acc[a]; }
    if (MU) {
      T G[9], D[9];
      continuous_blocks(c, qi, vi, G, D);
      for (int col = 0; col < 16; ++col) {
        for (int r = 0; r < 6; ++r) K[r * 16 + col] = A[(6 + r) * 16 + col];
        for (int r = 0; r < 3; ++r) {
          T gl = T(0), dw = T(0);
          for (int m = 0; m < 3; ++m) {
            gl += G[3 * r + m] * A[(3 + m) * 16 + col];
            dw += D[3 * r + m] * A[(9 + m) * 16 + col];
          }
          const T fu_l = (col >= 12) ? c.Bu[(6 + r) * 4 + (col - 12)] / c.dt : T(0);
          const T fu_w = (col >= 12) ? c.Bu[(9 + r) * 4 + (col - 12)] / c.dt : T(0);
          K[(6 + r) * 16 + col] = gl + fu_l;
          K[(9 + r) * 16 + col] = dw + fu_w;
        }
      }
      for (int e = 0; e < 192; ++e) SK[e] += coeffs[i] * K[e];
    }
    for (int e = 0; e < 12; ++e) xdot[e] += coeffs[i] * k[e];
  }
  // x_next = x (+) dt xdot
  T tau[6], tn[3], qn[4];
  for (int a = 0; a < 6; ++a) tau[a] = c.dt * xdot[a];
  se3_rplus(t, q, tau, tn, qn);
  if (MU) {
    ExpBlocks<T> eb;
    exp_blocks(tau, eb);
    T JK[96];
    blk6_mul(eb.Jr, eb.Qm, SK, JK);
    for (int r = 0; r < 6; ++r)
      for (int col = 0; col < 16; ++col) {
        T ad = T(0);
        if (col < 6) {
          const int br = r / 3, bc = col / 3, rr = r % 3, cc = col % 3;
          ad = (br == bc) ? eb.Rc[3 * rr + cc] : (br == 0 ? eb.SR[3 * rr + cc] : T(0));
        }
        MU[r * 16 + col] = ad + c.dt * JK[r * 16 + col];
        MU[(6 + r) * 16 + col] = ((col == 6 + r) ? T(1) : T(0)) + c.dt * SK[(6 + r) * 16 + col];
      }
  }
  for (int i = 0; i < 3; ++i) t[i] = tn[i];
  for (int i = 0; i < 4; ++i) q[i] = qn[i];
  for (int a = 0; a < 6; ++a) v[a] = v[a] + c.dt * xdot[6 + a];
}
// dynamics half of the linearisation for the Runge-Kutta step: the dense M at the head of the record (RecLayout.dense_m)
template <typename T, typename W>
QILQR_HD void linearize_dynamics_rk4(const ModelConsts<T> &c, const T *pt, W &w) {
  T t[3] = {pt[1], pt[2], pt[3]}, q[4] = {pt[5], pt[6], pt[7], pt[4]}, v[6];
  for (int i = 0; i < 6; ++i) v[i] = pt[8 + i];
  T MU[192];
  rk4_step(c, t, q, v, pt + 14, MU);
  for (int e = 0; e < 192; ++e) w.put(LIN_BLK + e, MU[e]);
}
// Linearisation of one knot in two independent halves (k_linearize runs them in different lanes).
// Dynamics: the six Jacobian blocks of quadrotor_model.cc:33-49, 84-119, 174-200, 266-276.
template <typename T, typename W>
QILQR_HD void linearize_dynamics(const ModelConsts<T> &c, const T *pt, W &w) {
  const T q[4] = {pt[5], pt[6], pt[7], pt[4]};
  const T *v = pt + 8;
  // ---- dynamics: tau = dt v ; E = Exp(tau) = (p, qe)
  T tau[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) tau[i] = c.dt * v[i];
  {
    T Jl[9], p[3], qe[4];
    so3_ljac_fast(tau + 3, Jl);
    mat3_vec(Jl, tau, p);
    so3_exp(tau + 3, qe);
    // Ad(E^-1) = [[Rc, hat(ti) Rc],[0, Rc]],  Rc = R(qe*), ti = -Rc p
    const T qc[4] = {-qe[0], -qe[1], -qe[2], qe[3]};
    T Rc[9], r[3], ti[3], S[9], SR[9];
    quat_to_R(qc, Rc);
    mat3_vec(Rc, p, r);
    ti[0] = -r[0]; ti[1] = -r[1]; ti[2] = -r[2];
    skew3(ti, S);
    mat3_mul(S, Rc, SR);
#pragma unroll
    for (int i = 0; i < 9; ++i) w.put(LIN_BLK + 0 + i, Rc[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) w.put(LIN_BLK + 9 + i, SR[i]);
    // dt * rjac(tau) = dt [[Jr, Q(-tau)],[0, Jr]],  Jr = Jl^T
    T nrho[3] = {-tau[0], -tau[1], -tau[2]}, nth[3] = {-tau[3], -tau[4], -tau[5]}, Qm[9];
    se3_fillQ_fast(nrho, nth, Qm);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) w.put(LIN_BLK + 18 + 3 * i + j, Jl[3 * j + i] * c.dt);
#pragma unroll
    for (int i = 0; i < 9; ++i) w.put(LIN_BLK + 27 + i, Qm[i] * c.dt);
  }
  {
    // d(lin acc)/d(rot) = -g hat(R^T e_z), scaled by dt
    T R[9];
    quat_to_R(q, R);
    const T rz[3] = {R[6], R[7], R[8]};
    T H[9];
    skew3(rz, H);
#pragma unroll
    for (int i = 0; i < 9; ++i) w.put(LIN_BLK + 36 + i, c.dt * (-c.g * H[i]));
    // d(ang acc)/d(omega) = -I^-1 (hat(w) I - hat(I w)); block = I + dt * that
    const T *om = v + 3;
    T Wh[9], WI[9], Iw[3], IwH[9], Jd[9], S[9];
    skew3(om, Wh);
    mat3_mul(Wh, c.inertia, WI);
    mat3_vec(c.inertia, om, Iw);
    skew3(Iw, IwH);
#pragma unroll
    for (int i = 0; i < 9; ++i) Jd[i] = WI[i] - IwH[i];
    mat3_mul(c.inertia_inv, Jd, S);
#pragma unroll
    for (int i = 0; i < 9; ++i) w.put(LIN_BLK + 45 + i, ((i % 4 == 0) ? T(1) : T(0)) + c.dt * (-S[i]));
  }
}

// Cost: value and differentials of cost.hh:36-61; returns the knot cost.
// LK = layout kind of the record, a compile-time choice (one kernel instantiation each, so that no weight
// is loaded on two sides of a run-time branch): 0 general Q, 1 symmetric Q, 2 symmetric Q with zero
// pose x velocity blocks; 3 (round 3): the record of 2 for a Q that is exactly DIAGONAL -- the reference's demo and every
// test of it use Q = diag -- where J^T Q is a row scaling: the same bits as 2 (the sums of 2 add exact zeros to these
// products), 190 fewer fp64 instructions and 100 fewer weight reads per knot.
QILQR_HD constexpr int layout_kind(const RecLayout &L) { return !L.sym ? 0 : (L.ur_zero ? 2 : 1); }
template <int LK, typename T, typename W>
QILQR_HD T linearize_cost(const T *Q, const T *R, const T *pt, const T *pd, W &w) {
  constexpr RecLayout L = make_layout(LK > 0, LK >= 2);
  constexpr bool DIAG = (LK == 3);
  // ---- cost: dx = x (-) x_d, J = blkdiag(Jri(tau_c), I6), Jri = [[a, -b],[0, a]] (3x3 blocks)
  T dx[12], du[4], sq[12], sr[4];
  RotScalars<T> rs;
  const T cost = knot_cost<(LK >= 2), T, DIAG>(Q, R, pt, pd, dx, du, sq, sr, &rs);
  T a[9], nb[9];  // Jri blocks: a = rjacinv of the rotation, nb = -a Q(-tau) a
  {
    T Li[9], Qm[9], aq[9];
    so3_ljacinv_with(dx + 3, rs.c, Li);  // (c, theta, sin, cos: as the Log inside knot_cost formed them)
    transpose3(Li, a);  // rjacinv = ljacinv^T
    T nrho[3] = {-dx[0], -dx[1], -dx[2]}, nth[3] = {-dx[3], -dx[4], -dx[5]};
    se3_fillQ_with(nrho, nth, rs, Qm);
    mat3_mul(a, Qm, aq);
    mat3_mul(aq, a, nb);
#pragma unroll
    for (int i = 0; i < 9; ++i) nb[i] = -nb[i];
  }
  // column j of Jri: rows 0..2 = a[:, j] (j < 3) or nb[:, j - 3]; rows 3..5 = 0 (j < 3) or a[:, j - 3].
  // The structural zeros are skipped below (they would only add exact zeros).
  // C_x = 2 (dx^T Q) J ; C_u = 2 du^T R
  {
    T gx[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      T s = T(0), v = T(0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        s += (T(2) * sq[r]) * a[3 * r + j];
        v += (T(2) * sq[r]) * nb[3 * r + j];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) v += (T(2) * sq[3 + r]) * a[3 * r + j];
      gx[j] = s;
      gx[3 + j] = v;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) w.put(L.off_g + j, gx[j]);
#pragma unroll
    for (int j = 6; j < 12; ++j) w.put(L.off_g + j, T(2) * sq[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) w.put(L.off_g + 12 + j, T(2) * sr[j]);
  }
  w.put(L.off_cost, cost);
  // C_xx = 2 (J^T Q) J, row by row (the order the record wants, and the association Eigen uses):
  //   wrow[k] = (J^T Q)[i][k] = sum_r J[r][i] Q[r][k],   C_xx[i][j] = 2 sum_k wrow[k] J[k][j]
  // with column c of Jri: rows 0..2 = a[:, c] (c < 3) or nb[:, c - 3]; rows 3..5 = 0 (c < 3) or a[:, c - 3].
  constexpr int NK = (LK >= 2) ? 6 : 12;  // columns of Q that matter for the rows i < 6
  auto jtq_row = [&](int i, T wrow[12]) {
    QILQR_REFETCH();
    if constexpr (DIAG) {
      // (J^T Q)[i][k] = J[k][i] Q[k][k]: column i of Jri scaled row by row
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const T qkk = Q[k * 12 + k];
        if (i < 3) wrow[k] = (k < 3) ? a[3 * k + i] * qkk : T(0);
        else wrow[k] = (k < 3) ? nb[3 * k + (i - 3)] * qkk : a[3 * (k - 3) + (i - 3)] * qkk;
      }
      QILQR_SCHED_FENCE();
      return;
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      T s = T(0);
      if (i < 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) s += a[3 * r + i] * Q[r * 12 + k];
      } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) s += nb[3 * r + (i - 3)] * Q[r * 12 + k];
#pragma unroll
        for (int r = 0; r < 3; ++r) s += a[3 * r + (i - 3)] * Q[(3 + r) * 12 + k];
      }
      wrow[k] = s;
    }
    QILQR_SCHED_FENCE();
  };
  auto times_jri = [&](const T *row, int j) {  // sum_k row[k] Jri[k][j], k < 6
    T s = T(0);
    if (j < 3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) s += row[k] * a[3 * k + j];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) s += row[k] * nb[3 * k + (j - 3)];
#pragma unroll
      for (int k = 0; k < 3; ++k) s += row[3 + k] * a[3 * k + (j - 3)];
    }
    return s;
  };
  if constexpr (LK == 0) {
    // dense 12x12, row-major: [[J^T Qpp J, J^T Qpv],[Qvp J, Qvv]]
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T wrow[12];
      jtq_row(i, wrow);
#pragma unroll
      for (int j = 0; j < 6; ++j) w.put(L.off_cxx + i * 12 + j, T(2) * times_jri(wrow, j));
#pragma unroll
      for (int j = 6; j < 12; ++j) w.put(L.off_cxx + i * 12 + j, T(2) * wrow[j]);
    }
#pragma unroll
    for (int i = 6; i < 12; ++i) {
      T qrow[12];
      QILQR_REFETCH();
#pragma unroll
      for (int k = 0; k < 12; ++k) qrow[k] = Q[i * 12 + k];
#pragma unroll
      for (int j = 0; j < 6; ++j) w.put(L.off_cxx + i * 12 + j, T(2) * times_jri(qrow, j));
#pragma unroll
      for (int j = 6; j < 12; ++j) w.put(L.off_cxx + i * 12 + j, T(2) * qrow[j]);
      QILQR_SCHED_FENCE();
    }
  } else {
    // symmetric Q: only the upper triangle of the 6x6 pose block and (unless it vanishes) the
    // pose x velocity block vary from knot to knot
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T wrow[12];
      jtq_row(i, wrow);
#pragma unroll
      for (int j = i; j < 6; ++j) w.put(L.off_cxx + symrow_index(LK == 1, i, j), T(2) * times_jri(wrow, j));
      if constexpr (LK == 1) {
#pragma unroll
        for (int j = 6; j < 12; ++j) w.put(L.off_cxx + symrow_index(true, i, j), T(2) * wrow[j]);
      }
    }
  }
  return cost;
}
template <typename T>
QILQR_HD void linearize_knot(const ModelConsts<T> &c, const RecLayout &L, const T *pt, const T *pd, T *rec) {
  PlainRecWriter<T> wd{rec};
  if (L.dense_m) linearize_dynamics_rk4(c, pt, wd);
  else linearize_dynamics(c, pt, wd);
  PlainRecWriter<T> w{rec + (L.dense_m ? LIN_M_DENSE - LIN_M_BLOCKS : 0)};  // the cost entries follow M wherever it ends
  switch (layout_kind(L)) {
    case 0: linearize_cost<0>(c.Q, c.R, pt, pd, w); break;
    case 1: linearize_cost<1>(c.Q, c.R, pt, pd, w); break;
    default: linearize_cost<2>(c.Q, c.R, pt, pd, w); break;
  }
}

// ----------------------------------------------------------------- rollout arithmetic
// The rollout is one serial chain per trajectory (state -> control -> next state), so its
// per-knot latency is the solver's critical path.  The functions below compute exactly the
// quantities of se3_rminus / se3_rplus but (a) with vector identities instead of 3x3 matrices
// (Jl^-1 t = t - th x t / 2 + c th x (th x t),  R(q) v = v + 2 w (u x v) + 2 u x (u x v)) and
// (b) with the analytic series of the four scalar functions involved when the relative rotation
// is moderate, which removes sqrt / atan2 / sincos / divisions from the chain.  manif's
// small-angle switch (theta^2 <= 1e-10) is kept as is.  All of it is the same function of the
// same inputs as the dense formulation, to rounding.
template <typename T>
QILQR_HD T poly8(const T c[8], T x) {  // Estrin, degree 7
  const T x2 = x * x, x4 = x2 * x2;
  const T p01 = c[0] + c[1] * x, p23 = c[2] + c[3] * x, p45 = c[4] + c[5] * x, p67 = c[6] + c[7] * x;
  return (p01 + p23 * x2) + (p45 + p67 * x2) * x4;
}

template <typename T>
struct Series {
  // in x = theta^2, valid for x <= EXP_MAX (truncation < 1e-19)
  static constexpr T EXP_MAX = T(0.25);
  static constexpr T cos_half[8] = {1.0, -0.125, 0.0026041666666666665, -2.170138888888889e-05,
                                    9.68812003968254e-08, -2.691144455467372e-10, 5.096864498991235e-13,
                                    -7.001187498614334e-16};
  static constexpr T sin_half_over[8] = {0.5, -0.020833333333333332, 0.00026041666666666666,
                                         -1.5500992063492063e-06, 5.382288910934745e-09,
                                         -1.2232474797578965e-11, 1.9603324996120133e-14,
                                         -2.333729166204778e-17};
  static constexpr T jac_a[8] = {0.5, -0.041666666666666664, 0.001388888888888889, -2.48015873015873e-05,
                                 2.755731922398589e-07, -2.08767569878681e-09, 1.1470745597729725e-11,
                                 -4.779477332387385e-14};  // (1 - cos th)/th^2
  static constexpr T jac_b[8] = {0.16666666666666666, -0.008333333333333333, 0.0001984126984126984,
                                 -2.7557319223985893e-06, 2.505210838544172e-08, -1.6059043836821613e-10,
                                 7.647163731819816e-13, -2.8114572543455206e-15};  // (th - sin th)/th^3
  // The same four functions as sixteen-term series, for EXP_MAX < x <= EXP2_MAX (round 6: k_rollout16's Exp; |dt omega| up to 3.46 rad per
  // step).  They are entire functions: the series converges everywhere, the first omitted term is below 1e-21 at x = 12, and the
  // alternating sum's rounding stays at 3e-16 absolute (checked against 50-digit sums: tests/test_device_math_on_host.py) -- what sqrt, two
  // sines, two cosines and three divisions give, at a tenth of their cost.  [0] cos(th/2), [1] sin(th/2)/th, [2] (1 - cos th)/th^2,
  // [3] (th - sin th)/th^3; coefficient k of [f] is (-1)^k over 4^k (2k)!, 2^(2k+1) (2k+1)!, (2k+2)!, (2k+3)!.
  static constexpr T EXP2_MAX = T(12.0);
  static constexpr T exp2[4][16] = {
      {1.0, -0.125, 0.0026041666666666665, -2.170138888888889e-05, 9.68812003968254e-08, -2.691144455467372e-10, 5.096864498991235e-13,
       -7.001187498614334e-16, 7.292903644389931e-19, -5.958254611429682e-22, 3.919904349624791e-25, -2.1211603623510776e-28,
       9.606704539633503e-32, -3.694886361397501e-35, 1.2218539554885916e-38, -3.511074584737332e-42},
      {0.5, -0.020833333333333332, 0.00026041666666666666, -1.5500992063492063e-06, 5.382288910934745e-09, -1.2232474797578965e-11,
       1.9603324996120133e-14, -2.333729166204778e-17, 2.1449716601146855e-20, -1.5679617398499164e-23, 9.333105594344741e-27,
       -4.6112181790240814e-30, 1.9213409079267006e-33, -6.842382150736113e-37, 2.106644750842399e-40, -5.66302352376989e-44},
      {0.5, -0.041666666666666664, 0.001388888888888889, -2.48015873015873e-05, 2.755731922398589e-07, -2.08767569878681e-09,
       1.1470745597729725e-11, -4.779477332387385e-14, 1.5619206968586225e-16, -4.110317623312165e-19, 8.896791392450574e-22,
       -1.6117375710961184e-24, 2.4795962632247976e-27, -3.279889237069838e-30, 3.7699876288159054e-33, -3.8003907548547434e-36},
      {0.16666666666666666, -0.008333333333333333, 0.0001984126984126984, -2.7557319223985893e-06, 2.505210838544172e-08,
       -1.6059043836821613e-10, 7.647163731819816e-13, -2.8114572543455206e-15, 8.22063524662433e-18, -1.9572941063391263e-20,
       3.868170170630684e-23, -6.446950284384474e-26, 9.183689863795546e-29, -1.1309962886447716e-31, 1.216125041553518e-34,
       -1.151633562077195e-37}};
  // Barfoot Q-block coefficients C = (1 - x/2 - cos th)/x^2 and D = (C - 3 (th - sin th - th^3/6)/th^5)/2
  // in x = theta^2 (B is jac_b), valid for x <= EXP_MAX
  static constexpr T fillq_C[8] = {-0.041666666666666664, 0.001388888888888889, -2.48015873015873e-05,
                                   2.755731922398589e-07, -2.08767569878681e-09, 1.1470745597729725e-11,
                                   -4.779477332387385e-14, 1.5619206968586225e-16};
  static constexpr T fillq_D[8] = {-0.008333333333333333, 0.0003968253968253968, -8.267195767195768e-06,
                                   1.0020843354176688e-07, -8.029521918410807e-10, 4.58829823909189e-12,
                                   -1.9680200780418645e-14, 6.576508197299464e-17};
  // asin(s)/s in y = s^2, valid for y <= LOG_MAX
  static constexpr T LOG_MAX = T(0.0625);
  static constexpr T asin_lo[8] = {1.0, 0.16666666666666666, 0.075, 0.044642857142857144,
                                   0.030381944444444444, 0.022372159090909092, 0.017352764423076924,
                                   0.01396484375};
  static constexpr T asin_hi[8] = {0.011551800896139705, 0.009761609529194078, 0.008390335809616815,
                                   0.0073125258735988454, 0.006447210311889649, 0.005740037670841924, 0.0, 0.0};
  // 1/th^2 - (1 + cos th)/(2 th sin th) in x = theta^2, valid for x <= JINV_MAX
  static constexpr T JINV_MAX = T(0.26);
  static constexpr T jinv_c[8] = {0.08333333333333333, 0.001388888888888889, 3.306878306878307e-05,
                                  8.267195767195768e-07, 2.08767569878681e-08, 5.284190138687493e-10,
                                  1.3382536530684679e-11, 3.3896802963225827e-13};
};

// R(q) v for a unit quaternion q = (x,y,z,w)
template <typename T>
QILQR_HD void quat_rotate(const T q[4], const T v[3], T o[3]) {
  T t[3], c[3];
  cross3(q, v, t);
  t[0] *= 2; t[1] *= 2; t[2] *= 2;
  cross3(q, t, c);
  o[0] = v[0] + q[3] * t[0] + c[0];
  o[1] = v[1] + q[3] * t[1] + c[1];
  o[2] = v[2] + q[3] * t[2] + c[2];
}

// tau = Log(X^-1 Y), same value as se3_rminus, in two parts so that the rollout can run them in
// two cooperating wavefronts: part 1 (pose -> td, theta, c) needs only poses; part 2 applies
// Jl^-1(theta) to td.
template <typename T, typename SR>
QILQR_HD void se3_rminus_part1(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T td[3], T th[3], T &c,
                               const SR &sr, RotScalars<T> *rs = nullptr) {
#if defined(__clang__)
#pragma clang fp contract(off)  // x (-) x must be exactly zero (cost_test.cc:27-39)
#endif
  const T qc[4] = {-qx[0], -qx[1], -qx[2], qx[3]};
  const T d[3] = {ty[0] - tx[0], ty[1] - tx[1], ty[2] - tx[2]};
  T qd[4];
  quat_rotate(qc, d, td);
  quat_mul(qc, qy, qd);
  const T n = qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3];
  if (fabs(n - T(1)) > Eps<T>::manif) {
    const T sc = T(2) / (T(1) + n);
    qd[0] *= sc; qd[1] *= sc; qd[2] *= sc; qd[3] *= sc;
  }
  const T s2 = qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2];
  // so3_log: theta = coeff * q_v
  T coeff;
  if (!(s2 > Eps<T>::manif)) {
    coeff = T(2);
  } else if (qd[3] > T(0) && s2 <= Series<T>::LOG_MAX) {
    const T y4 = (s2 * s2) * (s2 * s2);
    coeff = T(2) * (poly8(sr.asin_lo, s2) + poly8(sr.asin_hi, s2) * (y4 * y4));
  } else {
    const T s = sqrt(s2);
    const T w = qd[3];
    coeff = T(2) * ((w < T(0)) ? atan2(-s, -w) : atan2(s, w)) / s;
  }
  // so3_ljacinv(theta) t = t - theta x t / 2 + c theta x (theta x t), with its own switch on theta^2
  const T th2 = coeff * coeff * s2;
  if (!(th2 > Eps<T>::manif)) {
    c = T(0);
  } else if (th2 <= Series<T>::JINV_MAX) {
    c = poly8(sr.jinv_c, th2);
  } else {
    // manif's closed form, evaluated as manif does (it is ill-conditioned near theta = pi, where
    // only the same evaluation order reproduces the same digits)
    const T theta = sqrt(th2);
    const T cs = cos(theta), sn = sin(theta);
    c = T(1) / th2 - (T(1) + cs) / (T(2) * theta * sn);
    if (rs) { rs->theta = theta; rs->sn = sn; rs->cs = cs; }
  }
  if (rs) {
    rs->th2 = th2;
    rs->c = c;
    rs->trig = th2 > Series<T>::EXP_MAX;
    if (rs->trig && !(th2 > Series<T>::JINV_MAX)) {  // (the band between the two series' limits)
      rs->theta = sqrt(th2);
      rs->cs = cos(rs->theta);
      rs->sn = sin(rs->theta);
    }
  }
  th[0] = qd[0] * coeff; th[1] = qd[1] * coeff; th[2] = qd[2] * coeff;
}
template <typename T>
QILQR_HD void se3_rminus_part1(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T td[3], T th[3], T &c) {
  se3_rminus_part1(ty, qy, tx, qx, td, th, c, Series<T>());
}
template <typename T>
QILQR_HD void se3_rminus_part2(const T td[3], const T th[3], T c, T rho[3]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  T w1[3], w2[3];
  cross3(th, td, w1);
  cross3(th, w1, w2);
  rho[0] = td[0] - T(0.5) * w1[0] + c * w2[0];
  rho[1] = td[1] - T(0.5) * w1[1] + c * w2[1];
  rho[2] = td[2] - T(0.5) * w1[2] + c * w2[2];
}
template <typename T, typename SR>
QILQR_HD void se3_rminus_fast(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6], const SR &sr) {
  T td[3], c;
  se3_rminus_part1(ty, qy, tx, qx, td, tau + 3, c, sr);
  se3_rminus_part2(td, tau + 3, c, tau);
}
template <typename T>
QILQR_HD void se3_rminus_fast(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6]) {
  T td[3], c;
  se3_rminus_part1(ty, qy, tx, qx, td, tau + 3, c);
  se3_rminus_part2(td, tau + 3, c, tau);
}
template <typename T>
QILQR_HD void se3_rminus_fast(const T ty[3], const T qy[4], const T tx[3], const T qx[4], T tau[6], RotScalars<T> &rs) {
  T td[3], c;
  se3_rminus_part1(ty, qy, tx, qx, td, tau + 3, c, Series<T>(), &rs);
  se3_rminus_part2(td, tau + 3, c, tau);
}

// (t, q) <- (t, q) * Exp(tau), same value as se3_rplus
template <typename T, typename SR>
QILQR_HD void se3_rplus_fast(T t[3], T q[4], const T tau[6], const SR &sr) {
  const T *rho = tau, *th = tau + 3;
  const T th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  T ch, sh, a, b;
  if (!(th2 > Eps<T>::manif)) {
    ch = T(1); sh = T(0.5); a = T(0.5); b = T(0);
  } else if (th2 <= Series<T>::EXP_MAX) {
    ch = poly8(sr.cos_half, th2);
    sh = poly8(sr.sin_half_over, th2);
    a = poly8(sr.jac_a, th2);
    b = poly8(sr.jac_b, th2);
  } else {
    const T theta = sqrt(th2);
    const T ha = T(0.5) * theta;
    const T s = sin(ha);
    ch = cos(ha);
    sh = s / theta;
    a = (T(1) - cos(theta)) / th2;
    b = (theta - sin(theta)) / (th2 * theta);
  }
  const T qe[4] = {sh * th[0], sh * th[1], sh * th[2], ch};
  T w1[3], w2[3], p[3], Rp[3], qo[4];
  cross3(th, rho, w1);
  cross3(th, w1, w2);
  p[0] = rho[0] + a * w1[0] + b * w2[0];
  p[1] = rho[1] + a * w1[1] + b * w2[1];
  p[2] = rho[2] + a * w1[2] + b * w2[2];
  quat_rotate(q, p, Rp);
  quat_mul_fused(q, qe, qo);
  const T n = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
  if (fabs(n - T(1)) > Eps<T>::manif) {
    const T sc = T(2) / (T(1) + n);
    qo[0] *= sc; qo[1] *= sc; qo[2] *= sc; qo[3] *= sc;
  }
  t[0] += Rp[0]; t[1] += Rp[1]; t[2] += Rp[2];
  q[0] = qo[0]; q[1] = qo[1]; q[2] = qo[2]; q[3] = qo[3];
}

template <typename T>
QILQR_HD void se3_rplus_fast(T t[3], T q[4], const T tau[6]) {
  se3_rplus_fast(t, q, tau, Series<T>());
}
// The series coefficients the pose chain of the rollout needs, as values held in vector registers for
// the whole loop.  As compile-time constants they are 112 scalar registers' worth: the compiler parked
// them in spare VGPR lanes and fetched them back with two v_readlane each, every knot (a sixth of the
// pose wave's knot).  load() makes them opaque run-time values once.
template <typename T>
struct RolloutSeries {
  T cos_half[8], sin_half_over[8], jac_a[8], jac_b[8], asin_lo[8], asin_hi[8], jinv_c[8];
  QILQR_HD void load() {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      cos_half[k] = Series<T>::cos_half[k]; sin_half_over[k] = Series<T>::sin_half_over[k];
      jac_a[k] = Series<T>::jac_a[k]; jac_b[k] = Series<T>::jac_b[k];
      asin_lo[k] = Series<T>::asin_lo[k]; asin_hi[k] = Series<T>::asin_hi[k];
      jinv_c[k] = Series<T>::jinv_c[k];
      QILQR_PIN(cos_half[k]); QILQR_PIN(sin_half_over[k]); QILQR_PIN(jac_a[k]); QILQR_PIN(jac_b[k]);
      QILQR_PIN(asin_lo[k]); QILQR_PIN(asin_hi[k]); QILQR_PIN(jinv_c[k]);
    }
  }
};

// body acceleration without forming the whole rotation matrix (only R^T e_z is needed)
template <typename T>
QILQR_HD void body_acceleration_fast(const ModelConsts<T> &c, const T q[4], const T v[6], const T u[4], T acc[6]) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const T r6 = tz * x - ty * w, r7 = tz * y + tx * w, r8 = T(1) - (tx * x + ty * y);
  const T usum = ((u[0] + u[1]) + u[2]) + u[3];
  acc[0] = -c.g * r6;
  acc[1] = -c.g * r7;
  acc[2] = -c.g * r8 + usum / c.mass;
  T M[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    M[i] = c.arms[4 * i] * u[0] + c.arms[4 * i + 1] * u[1] + c.arms[4 * i + 2] * u[2] + c.arms[4 * i + 3] * u[3];
  const T *om = v + 3;
  T Iw[3], wIw[3], rhs[3];
  mat3_vec(c.inertia, om, Iw);
  cross3(om, Iw, wIw);
#pragma unroll
  for (int i = 0; i < 3; ++i) rhs[i] = M[i] - wIw[i];
  mat3_vec(c.inertia_inv, rhs, acc + 3);
}

// u = (u_i + alpha k) + K dx (ilqr.hh:158-161); K dx in three independent partial sums per row
template <typename T>
QILQR_HD void control_law(const T *pt, const T *g, T alpha, const T dx[12], T u[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    T k0 = T(0), k1 = T(0), k2 = T(0);
#pragma unroll
    for (int col = 0; col < 4; ++col) {
      k0 += g[4 + col * 4 + a] * dx[col];
      k1 += g[4 + (col + 4) * 4 + a] * dx[col + 4];
      k2 += g[4 + (col + 8) * 4 + a] * dx[col + 8];
    }
    u[a] = (pt[14 + a] + alpha * g[a]) + ((k0 + k1) + k2);
  }
}

// closed-loop rollout of one problem (ilqr.hh:149-172).  traj/gains/out point at this problem's
// first element (knot_base) in the TILED or plain layout.
template <bool TILED, typename T, int INTEG = 0>
QILQR_HD void rollout_problem(const ModelConsts<T> &c, const T *traj, const T *gains, T alpha,
                              T *out, int n) {
  T pt[18], g[52];
  load_knot<TILED>(traj, 0, 18, pt);
  T t[3] = {pt[1], pt[2], pt[3]};
  T q[4] = {pt[5], pt[6], pt[7], pt[4]};
  T v[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = pt[8 + i];
  RolloutSeries<T> sr;  // series coefficients in registers for the whole loop
  sr.load();
  for (int i = 0; i < n; ++i) {
    load_knot<TILED>(traj, i, 18, pt);
    load_knot<TILED>(gains, i, 52, g);
    // dx = state (-) x_i
    T dx[12];
    const T qi[4] = {pt[5], pt[6], pt[7], pt[4]};
    se3_rminus_fast(t, q, pt + 1, qi, dx, sr);
#pragma unroll
    for (int a = 0; a < 6; ++a) dx[6 + a] = v[a] - pt[8 + a];
    T u[4];
    control_law(pt, g, alpha, dx, u);
    const T o[18] = {pt[0], t[0], t[1], t[2], q[3], q[0], q[1], q[2], v[0], v[1], v[2], v[3], v[4], v[5],
                     u[0], u[1], u[2], u[3]};
#pragma unroll
    for (int e = 0; e < 18; ++e) out[knot_elem<TILED>(i, e, 18)] = o[e];
    if (INTEG == 1) {
      if (i + 1 < n) rk4_step(c, t, q, v, u, (T *)nullptr);
    } else if (i + 1 < n) {  // the reference's step after the last knot is computed and discarded
      T acc[6], tau[6];
      body_acceleration_fast(c, q, v, u, acc);
#pragma unroll
      for (int a = 0; a < 6; ++a) tau[a] = c.dt * v[a];  // pose integrates with the OLD velocity
      se3_rplus_fast(t, q, tau, sr);
#pragma unroll
      for (int a = 0; a < 6; ++a) v[a] = v[a] + c.dt * acc[a];
    }
  }
}

}  // namespace qilqr

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
