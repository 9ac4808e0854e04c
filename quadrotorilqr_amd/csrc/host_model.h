// host_model.h -- host-side construction of the device constants from the QuadrotorModel
// constructor arguments (quadrotor_model.cc:6-25) and the cost weights.  Shared by the C ABI
// (ilqr_capi.hip) and the CPU test harness (tests/host_harness.cpp).
#pragma once
#include <cmath>
#include <cstring>

#include "se3_math.h"

namespace qilqr {

// 3x3 Cholesky (lower) the way Eigen's LLT proceeds; false on a non-positive pivot
inline bool chol3(const double A[9], double L[9]) {
  std::memset(L, 0, sizeof(double) * 9);
  for (int k = 0; k < 3; ++k) {
    double x = A[k * 3 + k];
    for (int j = 0; j < k; ++j) x -= L[k * 3 + j] * L[k * 3 + j];
    if (!(x > 0.0)) return false;
    x = std::sqrt(x);
    L[k * 3 + k] = x;
    for (int i = k + 1; i < 3; ++i) {
      double s = A[i * 3 + k];
      for (int j = 0; j < k; ++j) s -= L[i * 3 + j] * L[k * 3 + j];
      L[i * 3 + k] = s / x;
    }
  }
  return true;
}
inline void chol3_solve(const double L[9], const double *B, int nrhs, double *X) {
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

// false: "Inertia matrix is not positive definite!" (quadrotor_model.cc:19-24: LLT failed or
// the matrix is not isApprox-symmetric, Eigen's default precision 1e-12)
inline bool make_model_consts(double mass, const double inertia[9], double arm, double ttr, double g,
                              const double *Q, const double *R, double dt, ModelConsts<double> *c) {
  double L[9];
  const bool pd = chol3(inertia, L);
  double d2 = 0, n2 = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double d = inertia[i * 3 + j] - inertia[j * 3 + i];
      d2 += d * d;
      n2 += inertia[i * 3 + j] * inertia[i * 3 + j];
    }
  if (!pd || !(d2 <= 1e-24 * n2)) return false;
  c->dt = dt;
  c->mass = mass;
  c->g = g;
  std::memcpy(c->inertia, inertia, sizeof(c->inertia));
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  chol3_solve(L, eye, 3, c->inertia_inv);
  const double arms[12] = {0, -arm, 0, arm, arm, 0.0, -arm, 0.0, -ttr, ttr, -ttr, ttr};  // quadrotor_model.cc:15-18
  std::memcpy(c->arms, arms, sizeof(arms));
  // constant control Jacobian J_u = J_rhs Jc_u (quadrotor_model.cc:45, 113-119, 272)
  std::memset(c->Bu, 0, sizeof(c->Bu));
  double S[12];
  chol3_solve(L, arms, 4, S);
  for (int j = 0; j < 4; ++j) c->Bu[8 * 4 + j] = dt * (1.0 / mass);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) c->Bu[(9 + i) * 4 + j] = dt * S[i * 4 + j];
  std::memcpy(c->Q, Q, sizeof(c->Q));
  std::memcpy(c->R, R, sizeof(c->R));
  return true;
}

}  // namespace qilqr
