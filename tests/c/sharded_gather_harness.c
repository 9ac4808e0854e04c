/* sharded_gather_harness.c -- a plain C host for qilqr_solve_batch_sharded_device (tests only).
 *
 * What a C or C++ host of the reference would do with include/quadrotor_ilqr.h and nothing else: build a batch of
 * hover problems with random starts, solve it (a) on one device, (b) sharded with the results gathered into the root
 * device's memory over the requested transport, compare the two bit for bit and print one JSON line with the transport
 * the handle reports and the exposed gather time.  No Python, no torch in the process.
 *
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/sharded_gather_harness.c \
 *       -Lquadrotorilqr_amd/lib -lquadrotor_ilqr -L/opt/rocm/lib -lamdhip64 -lm -o tests/c/sharded_gather_harness
 *   usage: sharded_gather_harness <transport: 0 auto | 1 rccl | 2 peer> <B> <n> <root> <dev0> [dev1 ...]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quadrotor_ilqr.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static double uniform(void) { /* splitmix64 -> [-1, 1) */
  uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (double)(z >> 11) / 4503599627370496.0 - 1.0;
}

#define CHECK_Q(expr)                                                          \
  do {                                                                         \
    int rc_ = (expr);                                                          \
    if (rc_ != QILQR_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, qilqr_last_error());       \
      return 2;                                                                \
    }                                                                          \
  } while (0)
#define CHECK_H(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));               \
      return 3;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s transport B n root dev0 [dev1 ...]\n", argv[0]);
    return 1;
  }
  const int transport = atoi(argv[1]), B = atoi(argv[2]), n = atoi(argv[3]), root = atoi(argv[4]);
  const int k = argc - 5;
  int32_t devices[64];
  for (int r = 0; r < k && r < 64; ++r) devices[r] = atoi(argv[5 + r]);

  /* model A of BASELINE.md section 3: the demo's quadrotor with a torque-to-thrust ratio, hover as the desired trajectory */
  qilqr_model model = {1.0, {1, 0, 0, 0, 1, 0, 0, 0, 1}, 1.0, 0.1, 9.81};
  qilqr_options opt = {0.5, 0.5, 100, 1e-12, 1e-12, 100.0, 0};
  double Q[144] = {0}, R[16] = {0};
  for (int i = 0; i < 12; ++i) Q[i * 12 + i] = i < 6 ? 100.0 : 1.0;
  for (int i = 0; i < 4; ++i) R[i * 4 + i] = 1.0;
  const double dt = 0.1, hover = model.mass_kg * model.g_mpss / 4.0;
  double *desired = calloc((size_t)n * 18, sizeof(double));
  double *init = calloc((size_t)B * n * 18, sizeof(double));
  for (int i = 0; i < n; ++i) {
    double *p = desired + (size_t)i * 18;
    p[0] = i * dt;
    p[4] = 1.0; /* identity quaternion (w, x, y, z) */
    for (int u = 0; u < 4; ++u) p[14 + u] = hover;
  }
  for (int b = 0; b < B; ++b) {
    double *t = init + (size_t)b * n * 18;
    memcpy(t, desired, sizeof(double) * n * 18);
    /* knot 0: a random pose within a metre and about a quarter turn of the target, random body velocity */
    double ax[3], nn = 0.0;
    for (int c = 0; c < 3; ++c) t[1 + c] = uniform();
    for (int c = 0; c < 3; ++c) { ax[c] = uniform(); nn += ax[c] * ax[c]; }
    nn = sqrt(nn) + 1e-300;
    const double ang = 0.7 * uniform();
    t[4] = cos(ang / 2);
    for (int c = 0; c < 3; ++c) t[5 + c] = sin(ang / 2) * ax[c] / nn;
    for (int c = 0; c < 6; ++c) t[8 + c] = 0.5 * uniform();
  }

  const size_t cnt = (size_t)B * n * 18;
  double *ref_traj = malloc(sizeof(double) * cnt), *ref_cost = malloc(sizeof(double) * B);
  int32_t *ref_int = malloc(sizeof(int32_t) * 4 * B);
  qilqr_solver *one = NULL;
  qilqr_device_config dc = {devices[root], 0, 2, 0, 0, 0, 0, 0, 0};
  CHECK_Q(qilqr_create(&model, Q, R, desired, n, dt, &opt, &dc, &one));
  CHECK_Q(qilqr_solve_batch(one, init, NULL, B, n, ref_traj, ref_cost, ref_int, ref_int + B, ref_int + 2 * B, ref_int + 3 * B));
  qilqr_destroy(one);

  qilqr_sharded *h = NULL;
  CHECK_Q(qilqr_sharded_create(&model, Q, R, desired, n, dt, &opt, NULL, devices, k, &h));
  CHECK_Q(qilqr_sharded_set_transport(h, transport));
  CHECK_H(hipSetDevice(devices[root]));
  double *d_traj, *d_cost;
  int32_t *d_int;
  CHECK_H(hipMalloc((void **)&d_traj, sizeof(double) * cnt));
  CHECK_H(hipMalloc((void **)&d_cost, sizeof(double) * B));
  CHECK_H(hipMalloc((void **)&d_int, sizeof(int32_t) * 4 * B));
  double gather_ms = -1.0, best_ms = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK_H(hipMemset(d_traj, 0xff, sizeof(double) * cnt));
    CHECK_Q(qilqr_solve_batch_sharded_device(h, init, NULL, B, n, root, d_traj, d_cost, d_int, d_int + B, d_int + 2 * B, d_int + 3 * B,
                                             &gather_ms));
    if (gather_ms < best_ms) best_ms = gather_ms;
  }
  double *got_traj = malloc(sizeof(double) * cnt), *got_cost = malloc(sizeof(double) * B);
  int32_t *got_int = malloc(sizeof(int32_t) * 4 * B);
  CHECK_H(hipMemcpy(got_traj, d_traj, sizeof(double) * cnt, hipMemcpyDeviceToHost));
  CHECK_H(hipMemcpy(got_cost, d_cost, sizeof(double) * B, hipMemcpyDeviceToHost));
  CHECK_H(hipMemcpy(got_int, d_int, sizeof(int32_t) * 4 * B, hipMemcpyDeviceToHost));
  const int same = memcmp(got_traj, ref_traj, sizeof(double) * cnt) == 0 && memcmp(got_cost, ref_cost, sizeof(double) * B) == 0 &&
                   memcmp(got_int, ref_int, sizeof(int32_t) * 4 * B) == 0;
  int converged = 0;
  for (int b = 0; b < B; ++b) converged += ref_int[b] == QILQR_STATUS_CONVERGED_EXPECTED || ref_int[b] == QILQR_STATUS_CONVERGED;
  printf("{\"transport\": \"%s\", \"shards\": %d, \"B\": %d, \"n\": %d, \"root\": %d, \"bit_identical\": %s, \"converged\": %d, "
         "\"gather_ms_best_of_3\": %.4f, \"gathered_bytes\": %zu}\n",
         qilqr_sharded_transport(h), k, B, n, root, same ? "true" : "false", converged, best_ms,
         sizeof(double) * cnt + (size_t)B * 24);
  qilqr_sharded_destroy(h);
  (void)hipFree(d_traj);
  (void)hipFree(d_cost);
  (void)hipFree(d_int);
  return same ? 0 : 4;
}
