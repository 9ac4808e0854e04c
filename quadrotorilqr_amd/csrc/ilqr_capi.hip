// ilqr_capi.hip -- host side of libquadrotor_ilqr.so: the C ABI of include/quadrotor_ilqr.h
// over the HIP kernels of ilqr_kernels.h.  C++ because the reference's host side is C++
// (src/quadrotor_ilqr_binding.cc, src/ilqr.hh); no exceptions cross the ABI.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define QILQR_NO_SIZED_MACROS  // (this file DEFINES the legacy symbols the macros stand in front of)
#include "../../include/quadrotor_ilqr.h"
#include "host_model.h"
#include "ilqr_kernels.h"

using namespace qilqr;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(QILQR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

enum Kind { K_BACKWARD = 0, K_ROLLOUT = 1, K_LINEARIZE = 2, K_OTHER = 3, K_SOLVE = 4, K_KINDS = 5 };

struct EventPair {
  hipEvent_t a, b;
  int kind;
};

}  // namespace

struct qilqr_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  ModelConsts<double> consts;
  SolveParams params;
  qilqr_options options;
  qilqr_device_config dev;
  int n_desired = 0;
  bool symmetric = false;  // Q == Q^T and R == R^T exactly: transpose-free backward kernel
  bool q_diag = false;     // Q exactly diagonal: the cost half of k_linearize scales rows instead of multiplying by Q (same bits)
  RecLayout layout;        // knot record layout chosen from the structure of Q
  void *d_desired = nullptr;    // shared desired trajectory, storage precision
  void *d_ctab = nullptr;       // constant operand table of k_backward, storage precision
  void *d_consts = nullptr;     // the model constants in device memory (k_linearize reads them where it uses them)
  bool f32 = false;             // mixed-precision mode (qilqr_device_config.precision == 1)
  int integrator = 0;           // 0 explicit Euler (the reference), 1 the Runge-Kutta extension (qilqr_set_integrator)
  ModelConsts<float> constsf;   // the model constants for the fp32 lane-local kernels
  // workspace
  long cap_B = 0, cap_n = 0;
  int hist_cap = 0;
  BatchState st{};
  std::vector<void *> allocs;
  int *h_counters = nullptr;  // pinned, 16 slots
  // pinned + mapped, 8 words per part written by k_linearize (BatchState::host_active): part 0 is the
  // whole batch on the main stream, parts 1..MAX_PARTS are sub-batches on their own streams
  unsigned long long *h_active = nullptr;
  unsigned long long *d_active = nullptr;  // the same memory as the device sees it
  static constexpr int MAX_PARTS = 8;
  hipStream_t part_stream[MAX_PARTS] = {};
  hipEvent_t part_done[MAX_PARTS] = {};
  hipEvent_t main_ready = nullptr;
  int *d_part_counters = nullptr;  // [MAX_PARTS][2][COUNT_WORDS]
  long total_B = 0;                // trajectories in flight on the device in this call (kernel choices go by it)
  bool round_captured = false;     // the round just enqueued was a k_round launch (it fills the single solve's debug ring itself)
  long live_hint = 0;              // trajectories known to be running in this call right now (0: unknown, take the batch): launch_backward
  double *io_aos = nullptr;         // device scratch in the plain [B][n][W] layout (W <= 52), for host I/O (lazy)
  size_t io_cap = 0;                // its capacity in doubles
  void *desired_tiled = nullptr;    // per-problem desired trajectories, tiled (allocated on first use)
  // device staging of the host-buffer batch entry point (qilqr_solve_batch), kept between calls, grow-only
  double *stage_traj = nullptr, *stage_des = nullptr, *stage_cost = nullptr;
  int *stage_int = nullptr;
  size_t stage_traj_cap = 0, stage_des_cap = 0, stage_B_cap = 0;
  // copy-back of the finished trajectories under the tail rounds of a host-buffer batch solve (EarlyOut below)
  void *early_out = nullptr;              // EarlyOut *, set by qilqr_solve_batch for the duration of its solve
  hipStream_t early_stream = nullptr;
  hipEvent_t early_evt = nullptr;
  hipEvent_t early_done = nullptr;  // the early part's copies have landed (the late finishers' rows are written behind it)
  int *d_early = nullptr;                 // [2 B]: early[B] | late_slot[B]
  size_t early_cap = 0;                   // B it was allocated for
  char *d_late = nullptr, *h_late = nullptr;  // compact rows of the trajectories that finished late: device block, pinned host block
  size_t late_bytes = 0;
  // ILQRDebug ring of the single-problem solve (k_debug_capture), device memory, grow-only
  double *dbg_trajs = nullptr, *dbg_cost = nullptr;
  int *dbg_seen = nullptr;
  size_t dbg_traj_cap = 0, dbg_cost_cap = 0;
  // profiling
  std::vector<EventPair> events;
  size_t events_used = 0;
  double prof_ms[K_KINDS] = {0, 0, 0, 0, 0};
  unsigned prof_seen[K_KINDS] = {0, 0, 0, 0, 0};  // launches of each kind seen by the sampler
  int prof_n[K_KINDS] = {0, 0, 0, 0, 0};
  int num_cus = 256;  // compute units of the device (grid of the persistent solve)
  // compaction of the live trajectories (k_compact_plan / k_compact_move): on for the duration of a device-resident batch
  // solve that qualifies (compaction_for_call), with the caller's result arrays for the trajectories that leave early
  bool compact = false;
  qilqr::CompactOut compact_out{};
  std::vector<long> plan_heads;  // where the plans of the last batch solve's (sub-)batches start in st.plan (qilqr_debug_compaction_moves)
};

namespace {

// roctx ranges (SURVEY.md section 5 "Tracing / profiling"; qilqr_device_config.profile bit 16): the host thread marks the call, every round it
// enqueues and -- for a batch on sub-batch streams -- every part's share of a round, so that a `rocprofv3 --marker-trace --kernel-trace` of a
// large solve reads as rounds of named parts instead of four streams of anonymous launches (the ranges bracket the ENQUEUE on the host; the
// kernels they enqueue carry their correlation).  libroctx64 is bound at first use from beside the HIP runtime; absent, the ranges are nothing.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char *name : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"}) {
      if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
  static Roctx &get() {
    static Roctx r;
    return r;
  }
};
struct RoctxRange {
  bool on = false;
  RoctxRange(const qilqr_solver *s, const char *what, long a = -1, long b = -1);
  ~RoctxRange() {
    if (on) (void)Roctx::get().pop();
  }
};

RoctxRange::RoctxRange(const qilqr_solver *s, const char *what, long a, long b) {
  if (!(s->dev.profile & 0x10000) || !Roctx::get().push) return;
  char buf[96];
  if (a >= 0 && b >= 0) std::snprintf(buf, sizeof buf, "qilqr %s %ld part %ld", what, a, b);
  else if (a >= 0) std::snprintf(buf, sizeof buf, "qilqr %s %ld", what, a);
  else std::snprintf(buf, sizeof buf, "qilqr %s", what);
  (void)Roctx::get().push(buf);
  on = true;
}
// Slot for the start/stop events of one launch, or null when this kind of kernel is not being timed.
EventPair *timing_slot(qilqr_solver *s, int kind) {
  const int mode = s->dev.profile & 0xff, stride = (s->dev.profile >> 8) & 0xff;
  if (!mode) return nullptr;
  const unsigned seen = s->prof_seen[kind]++;  // every launch of the kind since the last reset
  if (kind != K_SOLVE) {  // (the one launch of a persistent solve is always timed)
    if (mode == 1 && kind != K_BACKWARD && kind != K_ROLLOUT) return nullptr;
    if (mode == 3 && kind != K_BACKWARD) return nullptr;
    if (mode == 4 && kind != K_ROLLOUT) return nullptr;
    // sampling: every stride-th launch of a kind carries events (a timed dispatch costs the stream ~6 us)
    if (stride > 1 && (seen % stride) != 0) return nullptr;
  }
  if (s->events_used == s->events.size()) {
    EventPair e;
    if (hipEventCreate(&e.a) != hipSuccess) return nullptr;
    if (hipEventCreate(&e.b) != hipSuccess) {
      (void)hipEventDestroy(e.a);
      return nullptr;
    }
    s->events.push_back(e);
  }
  EventPair *ep = &s->events[s->events_used++];
  ep->kind = kind;
  return ep;
}
// Every kernel goes through here.  A timed launch hands its start/stop events to the dispatch itself
// (hipExtLaunchKernelGGL): the timestamps are the kernel's own begin and end, and no extra barrier
// packet enters the stream, so profiling does not stretch the round it measures.
template <typename... P, typename... Args>
void launch(qilqr_solver *s, int kind, void (*kernel)(P...), dim3 grid, dim3 block, Args... args) {
  EventPair *ep = timing_slot(s, kind);
  hipExtLaunchKernelGGL(kernel, grid, block, 0, s->stream, ep ? ep->a : nullptr, ep ? ep->b : nullptr, 0,
                        static_cast<P>(args)...);  // arguments converted to the kernel's own parameter types
}

void drain_events(qilqr_solver *s) {
  for (size_t i = 0; i < s->events_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->events[i].a, s->events[i].b) == hipSuccess) {
      s->prof_ms[s->events[i].kind] += ms;
      s->prof_n[s->events[i].kind] += 1;
    }
  }
  s->events_used = 0;
}

// After a stream has drained: did a kernel give up (BatchState::host_error)?  k_rollout16 ends a block whose wavefronts lost
// a hand-off instead of spinning for ever; the trajectories of that block are then not what the solve should have produced.
int device_error(qilqr_solver *s) {
  unsigned long long *w = s->h_active + 8 * (1 + qilqr_solver::MAX_PARTS);
  const unsigned long long v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
  if (!v) return QILQR_OK;
  __atomic_store_n(w, 0ull, __ATOMIC_RELEASE);
  if ((v >> 32) == 2)
    return fail(QILQR_ERR_HIP, "k_backward4: a hand-off between the wavefronts of block " + std::to_string((unsigned)v) +
                                   " never arrived (bounded spin ran out); its gains are void and the results of this call are invalid");
  return fail(QILQR_ERR_HIP, "k_rollout16: a hand-off between the wavefronts of block " + std::to_string((unsigned)v) +
                                 " never arrived (bounded spin ran out); its rollout was abandoned and the results of this call are invalid");
}

void free_workspace(qilqr_solver *s) {
  for (void *p : s->allocs) (void)hipFree(p);
  s->allocs.clear();
  s->cap_B = s->cap_n = 0;
  s->io_aos = nullptr;
  s->io_cap = 0;
  s->desired_tiled = nullptr;
}

template <typename T>
int dalloc(qilqr_solver *s, T **p, size_t count) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) return fail(QILQR_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  s->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return QILQR_OK;
}

int dalloc_s(qilqr_solver *s, void **p, size_t count) {  // count elements of the storage type
  char *q = nullptr;
  int rc = dalloc(s, &q, count * (s->f32 ? sizeof(float) : sizeof(double)));
  *p = q;
  return rc;
}

// number of ILQRIterDebug entries a solve can produce: the loop of ilqr.hh:58 runs for i = 0 .. while i < max_iters
// with max_iters a double, i.e. ceil(max_iters) times
inline int debug_capacity(double max_iters) { return (int)std::fmin(std::fmax(std::ceil(max_iters), 0.0), 1e6); }

int ensure_workspace(qilqr_solver *s, long B, long n) {
  const int want_hist = s->options.populate_debug ? debug_capacity(s->params.max_iters) : 0;
  if (B <= s->cap_B && n <= s->cap_n && want_hist <= s->hist_cap) return QILQR_OK;
  // grow only: alternating (B, n) shapes settle on the larger of each instead of reallocating on every call
  const long cB = B > s->cap_B ? B : s->cap_B, cn = n > s->cap_n ? n : s->cap_n;
  free_workspace(s);
  BatchState &st = s->st;
  st.layout = s->layout;
  st.ctab = s->d_ctab;
  int rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = dalloc_s(s, &st.traj[k], (size_t)tiled_count(cB, cn, 18)))) return rc;
    if ((rc = dalloc_s(s, &st.lin[k], (size_t)rec_count(cB, cn, s->layout.stride)))) return rc;
    if ((rc = dalloc(s, &st.knot_cost[k], (size_t)tiled_count(cB, cn, 1)))) return rc;
  }
  if ((rc = dalloc_s(s, &st.gains, (size_t)tiled_count(cB, cn, 52)))) return rc;
  s->io_aos = nullptr;  // host-I/O scratch: allocated on first use (ensure_io), the device-resident solve needs none
  s->io_cap = 0;
  s->desired_tiled = nullptr;
  if ((rc = dalloc(s, &st.cur, cB))) return rc;
  if ((rc = dalloc(s, &st.cost, cB))) return rc;
  if ((rc = dalloc(s, &st.prev_cost, cB))) return rc;
  if ((rc = dalloc(s, &st.terms, 2 * cB))) return rc;
  if ((rc = dalloc(s, &st.alpha, cB))) return rc;
  if ((rc = dalloc(s, &st.mu, cB))) return rc;
  if ((rc = dalloc(s, &st.trial, cB))) return rc;
  if ((rc = dalloc(s, &st.flags, cB))) return rc;
  if ((rc = dalloc(s, &st.status, cB))) return rc;
  if ((rc = dalloc(s, &st.iters, cB))) return rc;
  if ((rc = dalloc(s, &st.n_bwd, cB))) return rc;
  if ((rc = dalloc(s, &st.n_fwd, cB))) return rc;
  if ((rc = dalloc(s, &st.counters, 2 * COUNT_WORDS))) return rc;  // (two sets: k_round alternates between them)
  if ((rc = dalloc_s(s, &st.dump, 4 * cB))) return rc;
  if ((rc = dalloc(s, &st.orig, cB))) return rc;
  if ((rc = dalloc(s, &st.plan, (size_t)PLAN_HEAD * (qilqr_solver::MAX_PARTS + 2) + 4 * (size_t)cB))) return rc;  // (a part: head, B holes, B live slots, B / 2 pairs x 4)
  st.row0 = 0;
#if defined(QILQR_STAMPS) || defined(QILQR_ROUND_STAMPS)
  if ((rc = dalloc(s, &st.stamps, 8 * cB))) return rc;
#else
  st.stamps = nullptr;
#endif
  st.cost_hist = nullptr;
  st.hist_cap = 0;
  if (want_hist > 0) {
    if ((rc = dalloc(s, &st.cost_hist, (size_t)cB * want_hist))) return rc;
    st.hist_cap = want_hist;
  }
  s->hist_cap = want_hist;
  s->cap_B = cB;
  s->cap_n = cn;
  return QILQR_OK;
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
#ifndef QILQR_REGIME_B
#define QILQR_REGIME_B 4096
#endif
constexpr long REGIME_B = QILQR_REGIME_B;  // calls with more trajectories in flight take the kernels built for a full chip
constexpr long R16_MAX_B = REGIME_B;  // k_rollout16 for every rollout up to this many trajectories (launch_rollout)
// largest number of consecutive restarts lm_restart (kernels_common.h) can grant one iteration
inline double max_restarts(const SolveParams &p) {
  if (!(p.mu_init > 0.0) || !(p.mu_init <= p.mu_max)) return 0.0;
  return 1.0 + std::floor(std::log(p.mu_max / p.mu_init) / std::log(p.mu_factor));
}

// plain [B][n][W] fp64 (device) -> tiled, storage precision
// (zero_word: an int the same launch sets to zero -- the group queue of a persistent solve that follows)
int to_tiled(qilqr_solver *s, const double *d_plain, void *tiled, long B, long n, int W, int *zero_word = nullptr) {
  if (s->f32)
    launch(s, K_OTHER, k_retile<float>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * (W / 2) * TILE, 256)), dim3(256), d_plain, (double *)nullptr,
                       (float *)tiled, (float *)tiled, (const int *)nullptr, 0, (int)B, (int)n, W, 1, zero_word);
  else
    launch(s, K_OTHER, k_retile<double>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * (W / 2) * TILE, 256)), dim3(256), d_plain,
                       (double *)nullptr, (double *)tiled, (double *)tiled, (const int *)nullptr, 0, (int)B, (int)n, W, 1, zero_word);
  return QILQR_OK;
}
// tiled -> plain [B][n][W] fp64 (device); sel/flip choose between t0 and t1 per trajectory
int from_tiled(qilqr_solver *s, double *d_plain, void *t0, void *t1, const int *sel, int flip, long B, long n, int W) {
  if (s->f32)
    launch(s, K_OTHER, k_retile<float>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * (W / 2) * TILE, 256)), dim3(256), (const double *)nullptr,
                       d_plain, (float *)t0, (float *)t1, sel, flip, (int)B, (int)n, W, 0, (int *)nullptr);
  else
    launch(s, K_OTHER, k_retile<double>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * (W / 2) * TILE, 256)), dim3(256), (const double *)nullptr,
                       d_plain, (double *)t0, (double *)t1, sel, flip, (int)B, (int)n, W, 0, (int *)nullptr);
  return QILQR_OK;
}

// bind the desired trajectory (shared, or per problem: plain device array, re-tiled here) and reset
// the buffer selectors
bool use_persistent(const qilqr_solver *s, long B);
bool records_tiled(const qilqr_solver *s, long load_B, bool persistent);
int begin_batch(qilqr_solver *s, long B, long n, const double *d_desired_batch) {
  if (B <= 0 || n <= 0) return fail(QILQR_ERR_INVALID_ARG, "B and n must be positive");
  if (!d_desired_batch && n > s->n_desired)
    return fail(QILQR_ERR_LENGTH_MISMATCH, "trajectory longer than desired trajectory");
  HIP_TRY(hipSetDevice(s->device));
  int rc = ensure_workspace(s, B, n);
  if (rc) return rc;
  if (d_desired_batch) {
    if (!s->desired_tiled && (rc = dalloc_s(s, &s->desired_tiled, (size_t)tiled_count(s->cap_B, s->cap_n, 18)))) return rc;
    if ((rc = to_tiled(s, d_desired_batch, s->desired_tiled, B, n, 18))) return rc;
    s->st.desired = s->desired_tiled;
    s->st.desired_tiled = 1;
  } else {
    s->st.desired = s->d_desired;
    s->st.desired_tiled = 0;
  }
  s->total_B = B;
  s->live_hint = 0;  // (nothing known yet: launch_backward takes the batch)
  s->st.layout.tiled = records_tiled(s, B, use_persistent(s, B)) ? 1 : 0;
  launch(s, K_OTHER, k_begin, dim3(cdiv(B, 256)), dim3(256), s->st, (int)B);
  return QILQR_OK;
}
// the plain-layout device scratch of the host-buffer entry points, sized to what the call needs
int ensure_io(qilqr_solver *s, size_t count) {
  if (count <= s->io_cap) return QILQR_OK;
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->io_aos) {
    for (auto it = s->allocs.begin(); it != s->allocs.end(); ++it)
      if (*it == (void *)s->io_aos) {
        s->allocs.erase(it);
        break;
      }
    (void)hipFree(s->io_aos);
    s->io_aos = nullptr;
    s->io_cap = 0;
  }
  int rc = dalloc(s, &s->io_aos, count);
  if (rc) return rc;
  s->io_cap = count;
  return QILQR_OK;
}
// host plain array -> device tiled buffer through the io scratch
int upload_tiled(qilqr_solver *s, const double *h_plain, void *tiled, long B, long n, int W) {
  int rc0 = ensure_io(s, (size_t)B * n * W);
  if (rc0) return rc0;
  HIP_TRY(hipMemcpyAsync(s->io_aos, h_plain, sizeof(double) * (size_t)B * n * W, hipMemcpyHostToDevice, s->stream));
  return to_tiled(s, s->io_aos, tiled, B, n, W);
}
int download_tiled(qilqr_solver *s, double *h_plain, void *t0, void *t1, const int *sel, int flip, long B, long n,
                   int W) {
  int rc = ensure_io(s, (size_t)B * n * W);
  if (rc) return rc;
  if ((rc = from_tiled(s, s->io_aos, t0, t1, sel, flip, B, n, W))) return rc;
  HIP_TRY(hipMemcpyAsync(h_plain, s->io_aos, sizeof(double) * (size_t)B * n * W, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return QILQR_OK;
}

int launch_linearize(qilqr_solver *s, long B, long n, int which, int need_flag, int round = -1) {
  const dim3 grid(cdiv(2 * ((B + 63) / 64) * 64 * n, QILQR_LIN_BLOCK));  // dynamics half + cost half
#define QILQR_LAUNCH_LIN(S, LK, TILED, CONSTS, DCONSTS) \
  launch(s, K_LINEARIZE, (k_linearize<S, LK, 0, TILED>), grid, dim3(QILQR_LIN_BLOCK), CONSTS, DCONSTS, s->st, (int)B, (int)n, which, need_flag, round)
#define QILQR_LAUNCH_LIN_RK4(LK) \
  launch(s, K_LINEARIZE, (k_linearize<double, LK, 1, false>), grid, dim3(QILQR_LIN_BLOCK), s->consts, (const ModelConsts<double> *)s->d_consts, s->st, (int)B, (int)n, which, need_flag, round)
  if (s->integrator == 1) {  // the Runge-Kutta extension: dense M at the head of the record, fp64 only, plain placement
    switch (layout_kind(s->layout)) {
      case 0: QILQR_LAUNCH_LIN_RK4(0); break;
      case 1: QILQR_LAUNCH_LIN_RK4(1); break;
      default: QILQR_LAUNCH_LIN_RK4(2); break;
    }
    return QILQR_OK;
  }
  // (the placement of the records, s->st.layout.tiled, was chosen with the call's backward kernel: records_tiled)
  if (layout_kind(s->layout) == 2 && s->q_diag && !s->f32) {
    // diagonal Q: the record of kind 2, cheaper arithmetic, the same bits in fp64 (tests/test_gpu_parity.py).  (Not in the
    // mixed mode: there the two instantiations differ in the last fp32 bit of a third of the knot costs -- the compiler
    // contracts the single-precision expressions differently -- and "the same results whatever the weights' structure" is
    // worth more than 1 % of k_linearize.)
    if (s->st.layout.tiled) QILQR_LAUNCH_LIN(double, 3, true, s->consts, (const ModelConsts<double> *)s->d_consts);
    else QILQR_LAUNCH_LIN(double, 3, false, s->consts, (const ModelConsts<double> *)s->d_consts);
    return QILQR_OK;
  }
  switch (layout_kind(s->layout) + (s->f32 ? 3 : 0) + (s->st.layout.tiled ? 6 : 0)) {
    case 0: QILQR_LAUNCH_LIN(double, 0, false, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 1: QILQR_LAUNCH_LIN(double, 1, false, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 2: QILQR_LAUNCH_LIN(double, 2, false, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 3: QILQR_LAUNCH_LIN(float, 0, false, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
    case 4: QILQR_LAUNCH_LIN(float, 1, false, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
    case 5: QILQR_LAUNCH_LIN(float, 2, false, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
    case 6: QILQR_LAUNCH_LIN(double, 0, true, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 7: QILQR_LAUNCH_LIN(double, 1, true, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 8: QILQR_LAUNCH_LIN(double, 2, true, s->consts, (const ModelConsts<double> *)s->d_consts); break;
    case 9: QILQR_LAUNCH_LIN(float, 0, true, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
    case 10: QILQR_LAUNCH_LIN(float, 1, true, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
    default: QILQR_LAUNCH_LIN(float, 2, true, s->constsf, (const ModelConsts<float> *)s->d_consts); break;
  }
#undef QILQR_LAUNCH_LIN
#undef QILQR_LAUNCH_LIN_RK4
  return QILQR_OK;
}
// Which backward kernel a call with `load_B` trajectories in flight takes (symmetric weights), by how many trajectories share
// the chip's 1024 SIMDs:
//   up to 4096: k_backward4<.., FUSED>: four wavefronts that each carry the matrix AND the gradient recursion of a trajectory,
//               and one loader wavefront, per four trajectories (the wavefronts are bound by latencies, the gradient's 40
//               instructions ride along: +0.3 to +1.7 % of a whole solve against the form below, profiles/r03_ab_backward.txt)
//   beyond:     k_backward4: four matrix wavefronts, ONE gradient wavefront and the loader per four trajectories, knot loop
//               unrolled (four blocks per CU: the SIMDs are bound by what their wavefronts issue, and one gradient wavefront
//               for four trajectories issues a quarter: 426k against 408k solves/s at 8192)
// Until round 4 the one-wavefront kernel (k_backward<true>) took over above 8192 trajectories: a block per trajectory wastes
// nothing on finished neighbours.  With the live trajectories compacted (k_compact_*) the blocks of four are full, and the
// six-wavefront form is ahead at every size measured (profiles/r04_compaction.txt: 12288: 554k against 464k solves/s,
// 16384: 593k / 512k, 65536: 654k / 587k); the one-wavefront kernel stays for force_general = 2.
// k_backward2 (a matrix and a gradient wavefront per trajectory) was the choice below 512 trajectories in rounds 1 and 2; it
// wins nowhere by more than 2 % and lives in the diagnostics build (force_general = 3 there).
// The Runge-Kutta extension and non-symmetric weights take the one-wavefront kernel at every size.
#ifndef QILQR_GFAC_MIN_LIVE
#define QILQR_GFAC_MIN_LIVE 3072
#endif
constexpr long GFAC_MIN_LIVE = QILQR_GFAC_MIN_LIVE;  // running trajectories from which the gradient wavefront factors Q_uu (launch_backward)
enum BackwardKind { BW_FOUR, BW_TWO, BW_ONE, BW_FUSED };
BackwardKind backward_kind(const qilqr_solver *s, long load_B) {
  if (s->integrator == 1 || !s->symmetric) return BW_ONE;
#ifdef QILQR_WITH_BACKWARD2
  if (s->dev.force_general == 3) return BW_TWO;
#endif
  if (s->dev.force_general == 5 || (s->dev.force_general == 0 && load_B <= REGIME_B)) return BW_FUSED;
  if (s->dev.force_general != 2) return BW_FOUR;
  return BW_ONE;
}
// The knot records are placed for their reader (se3_math.h, rec_base): tiled for the kernels that stage them through LDS
// (k_backward4, k_backward2, k_solve4), plain for the one-wavefront kernel.  Decided once per call, with the batch size
// the kernel choice goes by.
bool records_tiled(const qilqr_solver *s, long load_B, bool persistent) {
  return persistent || backward_kind(s, load_B) != BW_ONE;
}
int launch_backward(qilqr_solver *s, long B, long n, int force) {
#define QILQR_LAUNCH_BWD(SYM, S)                                                                              \
  launch(s, K_BACKWARD, (k_backward<SYM, S>), dim3((unsigned)B), dim3(64), s->consts, s->params, s->st, \
                     (int)B, (int)n, force)
  const long load_B = std::max(B, s->total_B);
  // (the records were linearised in the placement this choice reads: begin_batch sets st.layout.tiled from the same function;
  // the one-wavefront kernel addresses its operands through rec_elem and reads either)
  BackwardKind kind = s->st.layout.tiled ? backward_kind(s, load_B) : BW_ONE;
  // `live` = the trajectories known to be running on the device in this call (every sub-batch stream's last count; the batch while nothing is known)
  const long live = s->live_hint > 0 ? s->live_hint : load_B;
  // Since round 6 the fused and the six-wavefront forms give the same bits, so a batch of up to 4096 trajectories takes the six-wavefront form
  // (Q_uu factored by the gradient wavefront, four blocks per CU) for the launches in which most of it is still running -- every trajectory live,
  // per launch: 4096: 248 against 293 us, 3072: 177 / 184, 2048: 123 / 127, 1024: 86 / 74 -- and the fused form from there on.
  if (kind == BW_FUSED && s->dev.force_general == 0 && live >= GFAC_MIN_LIVE) kind = BW_FOUR;
  if (kind == BW_FUSED) {
    // four matrix-and-gradient wavefronts + one loader wavefront per four trajectories, no block barrier in the knot loop
    // (one register budget: the pipelined knot carries the previous knot's tail and does not fit 80 registers)
#define QILQR_LAUNCH_FUSED(S) \
  launch(s, K_BACKWARD, (k_backward4<S, 5, true, true>), dim3(cdiv(B, 4)), dim3(320), s->consts, s->params, s->st, (int)B, (int)n, force)
    if (s->f32) QILQR_LAUNCH_FUSED(float);
    else QILQR_LAUNCH_FUSED(double);
#undef QILQR_LAUNCH_FUSED
  } else if (kind == BW_FOUR) {
    // four matrix wavefronts + one gradient wavefront + one loader wavefront per four trajectories
    // (register budget by how many blocks the chip has to hold: see k_backward4)
#ifdef QILQR_FORCE_MANY  // (experiment: the four-blocks-per-CU register budget and the unrolled knot loop at every size)
    const bool many = true;
#else
    const bool many = load_B > REGIME_B;
#endif
    // Who factors Q_uu (round 6; the same bits either way, backward4_kernel.h): the gradient wavefront when the chip is saturated -- a SIMD
    // is then bound by what its wavefronts issue, and one instruction stream factors four trajectories' Q_uu instead of four (a launch with
    // every trajectory live, MI355X, N = 100: B = 65536 3807 -> 3515 us, 8192 509 -> 484) -- and the matrix wavefronts when a launch's
    // wavefronts are alone on their SIMDs and its time is the chain of one knot's dependent instructions (B = 64: 75.3 against 85.3 us;
    // level at 2048).
    const bool gfac = s->dev.force_general == 7 || (s->dev.force_general != 8 && (many || s->dev.force_general == 0) && live >= GFAC_MIN_LIVE);
    if (gfac && s->f32)
      launch(s, K_BACKWARD, (k_backward4<float, 6, false, false, true>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
    else if (gfac)
      launch(s, K_BACKWARD, (k_backward4<double, 6, false, false, true>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
    else if (s->f32 && many)
      launch(s, K_BACKWARD, (k_backward4<float, 6>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
    else if (s->f32)
      launch(s, K_BACKWARD, (k_backward4<float, 5>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
    else if (many)
      launch(s, K_BACKWARD, (k_backward4<double, 6>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
    else
      launch(s, K_BACKWARD, (k_backward4<double, 5>), dim3(cdiv(B, 4)), dim3(384), s->consts, s->params, s->st, (int)B, (int)n, force);
#ifdef QILQR_WITH_BACKWARD2
  } else if (kind == BW_TWO) {
    // two cooperating wavefronts per trajectory (matrix recursion / gradient recursion + operand streaming)
    if (s->f32)
      launch(s, K_BACKWARD, k_backward2<float>, dim3((unsigned)B), dim3(128), s->consts, s->params, s->st, (int)B, (int)n,
             force);
    else
      launch(s, K_BACKWARD, k_backward2<double>, dim3((unsigned)B), dim3(128), s->consts, s->params, s->st, (int)B,
             (int)n, force);
#endif
  } else if (s->symmetric) {
    if (s->f32) QILQR_LAUNCH_BWD(true, float);
    else QILQR_LAUNCH_BWD(true, double);
  } else {
    if (s->f32) QILQR_LAUNCH_BWD(false, float);
    else QILQR_LAUNCH_BWD(false, double);
  }
#undef QILQR_LAUNCH_BWD
  return QILQR_OK;
}
#ifndef QILQR_ROLLOUT16_FROM
#define QILQR_ROLLOUT16_FROM 16
#endif
constexpr long ROLLOUT16_FROM = QILQR_ROLLOUT16_FROM;
// ordinal: which rollout of its solve this is for every trajectory that takes part (a running trajectory rolls out exactly once per round, so
// the k-th rollout of ANY problem happens in round k of ANY call: a property of the problem, not of the batch); -1: the stand-alone entry points
int launch_rollout(qilqr_solver *s, long B, long n, int need_flag, long ordinal = -1) {
  // Which rollout kernel, by how many trajectories share the chip (qilqr_device_config.single_wave_rollout):
  //   k_rollout16  sixteen lanes per trajectory, four trajectories per block: the shortest chain per trajectory and a
  //                block on every CU from 1024 trajectories on; up to R16_MAX_B trajectories
  //   k_rollout3   a lane per trajectory, three cooperating wavefronts per 64 trajectories: beyond
  //   k_rollout    a lane per trajectory, one wavefront (the Runge-Kutta extension; forced).  It was the choice above 16384
  //                trajectories until the live trajectories were compacted: with full wavefronts k_rollout3 is ahead there too
  //                (65536: 698k against 655k solves/s, 16384: 596k / 543k, profiles/r04_compaction.txt)
  const long load_B = std::max(B, s->total_B);
  const int choice = s->dev.single_wave_rollout;
  if (s->integrator == 1) {  // the Runge-Kutta extension: the lane-per-trajectory kernel only
    launch(s, K_ROLLOUT, (k_rollout<double, 1>), dim3(cdiv(B, 64)), dim3(64), s->consts, s->st, (int)B, (int)n, need_flag);
  } else if (choice == 1) {  // (a forced choice is honoured at every batch size; until round 4 also the choice above 16384)
    if (s->f32)
      launch(s, K_ROLLOUT, (k_rollout<float, 0>), dim3(cdiv(B, 64)), dim3(64), s->constsf, s->st, (int)B, (int)n, need_flag);
    else
      launch(s, K_ROLLOUT, (k_rollout<double, 0>), dim3(cdiv(B, 64)), dim3(64), s->consts, s->st, (int)B, (int)n, need_flag);
  } else if (choice == 3 || (choice == 0 && load_B <= R16_MAX_B) || (choice == 0 && ordinal >= ROLLOUT16_FROM)) {
    if (s->f32)
      launch(s, K_ROLLOUT, k_rollout16<float>, dim3(cdiv(B, 4)), dim3(192), s->consts, s->st, (int)B, (int)n, need_flag);
    else
      launch(s, K_ROLLOUT, k_rollout16<double>, dim3(cdiv(B, 4)), dim3(192), s->consts, s->st, (int)B, (int)n, need_flag);
  } else {
    if (s->f32)
      launch(s, K_ROLLOUT, k_rollout3<float>, dim3(cdiv(B, 64)), dim3(192), s->constsf, s->st, (int)B, (int)n, need_flag);
    else
      launch(s, K_ROLLOUT, k_rollout3<double>, dim3(cdiv(B, 64)), dim3(192), s->consts, s->st, (int)B, (int)n, need_flag);
  }
  return QILQR_OK;
}
// k_backward_rollout (round_kernels.h): the backward pass and the rollout of a round in one launch, when every block of four
// trajectories has a CU to itself (the rollout's register budget allows one block per CU) and the round's kernels are the
// fused k_backward4 and k_rollout16 anyway.  qilqr_device_config.round_launch = 1 keeps them apart (A/B).
// (the kernels of the round are the two the combined launch stands for: everything but the room on the chip)
// (total_B / tiled: the call's batch in flight and its record placement -- the solver's own during a call, the caller's for qilqr_describe,
// which asks about a batch without touching the handle's state)
bool fuse_kinds(const qilqr_solver *s, long B, long total_B, bool tiled) {
  const bool off = s->dev.round_launch == 1;  // (qilqr_device_config.round_launch: three launches per round, A/B)
  const long load_B = std::max(B, total_B);
  if (off || s->integrator != 0 || !s->symmetric || !tiled) return false;
  // (force_general = 8 with the combined launch: k_round with the six-wavefront backward pass in EVERY launch -- tests, A/B)
  if (!(s->dev.force_general == 0 || s->dev.force_general == 5 || s->dev.force_general == 8)) return false;
  if (s->dev.force_general != 8 && backward_kind(s, load_B) != BW_FUSED) return false;
  if (s->dev.force_general == 8 && (s->dev.round_launch != 0 || s->f32)) return false;  // (only k_round has the form: not k_backward_rollout)
  if (!(s->dev.single_wave_rollout == 0 || s->dev.single_wave_rollout == 3) || load_B > R16_MAX_B) return false;
  return true;
}
bool fuse_kinds(const qilqr_solver *s, long B) { return fuse_kinds(s, B, s->total_B, s->st.layout.tiled != 0); }
bool fuse_backward_rollout(const qilqr_solver *s, long B, long total_B, bool tiled) {
  return fuse_kinds(s, B, total_B, tiled) && cdiv(std::max(B, total_B), 4) <= (unsigned)s->num_cus;
}
bool fuse_backward_rollout(const qilqr_solver *s, long B) { return fuse_backward_rollout(s, B, s->total_B, s->st.layout.tiled != 0); }
// Batch solves in flight on a device, over all the handles of the process.  The combined kernel takes a whole CU per block of
// four trajectories (the rollout's register budget): alone on the chip that is +3 to +4 % of a solve, beside other solves'
// kernels it is in their way -- three handles in flight: 306 000-311 000 solves/s with it, 340 000 without.  A solve that
// finds another one in flight on its device therefore launches the two kernels apart (same bits either way).
// The choice is made again for every round a solve enqueues (a solve that is joined by another one changes over at its next round;
// the count covers the window in which a solve's host loop enqueues rounds -- the last `sync_every` rounds of a call that
// returns before its stream has drained are outside it).
constexpr int MAX_TRACKED_DEVICES = 64;  // (HIP ordinals of one process; a node has 8)
std::atomic<int> g_solves_in_flight[MAX_TRACKED_DEVICES];
struct InFlight {
  std::atomic<int> &n;
  explicit InFlight(int device) : n(g_solves_in_flight[(unsigned)device % MAX_TRACKED_DEVICES]) { n.fetch_add(1, std::memory_order_relaxed); }
  ~InFlight() { n.fetch_sub(1, std::memory_order_relaxed); }
  // (always: qilqr_device_config.fuse_in_flight = 1 keeps the combined launches beside other solves -- diagnostic)
  bool alone(bool always = false) const { return always || n.load(std::memory_order_relaxed) == 1; }
};
// k_round (round_kernels.h): the combined launch and the linearisation of its candidates in one.  fp64 storage only (the mixed mode keeps
// the two launches).  The round's counts go into the counter set of its parity; the launch publishes the round before it.
bool round_kernel_ok(const qilqr_solver *s) { return s->dev.round_launch == 0 && !s->f32; }
// rounds per launch of k_round where a launch may hold several (qilqr_device_config.rounds_per_launch = 1, 2 or 4: A/B; 0 = 4)
int rounds_per_launch(const qilqr_solver *s) {
  const int v = s->dev.rounds_per_launch;
  return (v == 1 || v == 2) ? v : 4;
}
int launch_round(qilqr_solver *s, long B, long n, long round, bool publish_prev, int rounds, bool six = false) {
  const ModelConsts<double> *cp = (const ModelConsts<double> *)s->d_consts;
  if (rounds == 2) six = false;  // (the six-wavefront form is instantiated for launches of one and of four rounds)
  const dim3 grid(cdiv(B, 4)), block(six ? 384 : 320);
  BatchState st = s->st;
  int *base = s->st.counters;
  st.counters = base + (round & 1) * COUNT_WORDS;
  int *prev = base + ((round + 1) & 1) * COUNT_WORDS;
  const int prev_round = publish_prev ? (int)((round - 1) & 0x3fffffff) : -1;
  const int lk = (s->q_diag && layout_kind(s->layout) == 2) ? 3 : (layout_kind(s->layout) == 2 ? 2 : 1);
#define QILQR_LAUNCH_ROUND(LK, R) launch(s, K_BACKWARD, (k_round<LK, R>), grid, block, s->consts, cp, s->params, st, (int)B, (int)n, prev, prev_round)
#define QILQR_LAUNCH_ROUND6(LK, R) launch(s, K_BACKWARD, (k_round<LK, R, true>), grid, block, s->consts, cp, s->params, st, (int)B, (int)n, prev, prev_round)
  if (six && rounds == 4) {
    if (lk == 3) QILQR_LAUNCH_ROUND6(3, 4);
    else if (lk == 2) QILQR_LAUNCH_ROUND6(2, 4);
    else QILQR_LAUNCH_ROUND6(1, 4);
  } else if (six) {
    if (lk == 3) QILQR_LAUNCH_ROUND6(3, 1);
    else if (lk == 2) QILQR_LAUNCH_ROUND6(2, 1);
    else QILQR_LAUNCH_ROUND6(1, 1);
  } else if (rounds == 4) {
    if (lk == 3) QILQR_LAUNCH_ROUND(3, 4);
    else if (lk == 2) QILQR_LAUNCH_ROUND(2, 4);
    else QILQR_LAUNCH_ROUND(1, 4);
  } else if (rounds == 2) {
    if (lk == 3) QILQR_LAUNCH_ROUND(3, 2);
    else if (lk == 2) QILQR_LAUNCH_ROUND(2, 2);
    else QILQR_LAUNCH_ROUND(1, 2);
  } else {
    if (lk == 3) QILQR_LAUNCH_ROUND(3, 1);
    else if (lk == 2) QILQR_LAUNCH_ROUND(2, 1);
    else QILQR_LAUNCH_ROUND(1, 1);
  }
#undef QILQR_LAUNCH_ROUND
#undef QILQR_LAUNCH_ROUND6
  return QILQR_OK;
}
int launch_backward_rollout(qilqr_solver *s, long B, long n) {
  if (s->f32)
    launch(s, K_BACKWARD, k_backward_rollout<float>, dim3(cdiv(B, 4)), dim3(320), s->consts, s->params, s->st, (int)B, (int)n);
  else
    launch(s, K_BACKWARD, k_backward_rollout<double>, dim3(cdiv(B, 4)), dim3(320), s->consts, s->params, s->st, (int)B, (int)n);
  return QILQR_OK;
}
int launch_accept(qilqr_solver *s, long B, long n, int ls_only) {
  launch(s, K_OTHER, k_accept, dim3(cdiv(B, 64)), dim3(64), s->params, s->st, (int)B, (int)n,
                     ls_only);
  return QILQR_OK;
}

// ---- compaction of the live trajectories (bookkeeping_kernels.h, k_compact_plan): between a round's backward pass and its rollout.
// Worth its two launches while the live trajectories fill more blocks than the device runs side by side; below
// COMPACT_STOP running trajectories every kernel of a round is a lone dependent chain whatever the slots are.
#ifndef QILQR_COMPACT_STOP
#define QILQR_COMPACT_STOP 512
#endif
constexpr unsigned COMPACT_STOP = QILQR_COMPACT_STOP;
// Automatic (qilqr_device_config.compaction = 0): whenever the round's backward pass is a k_backward4 (blocks of four trajectories)
// and not part of the combined launch of B <= 1024 -- measured, one configuration per process (profiles/r04_compaction.txt): 1280:
// +5 %, 2048: +5.5 %, 3072: +13 %, 4096: +8 %, 8192: +6 %; with the one-wavefront backward kernel (general weights, the Runge-Kutta
// extension, force_general = 2), whose blocks hold one trajectory, it gains nothing (12288-32768: -2 to +1 %) and stays off.
inline unsigned compact_stop(const qilqr_solver *s) { return s->dev.compaction == 1 ? 0u : COMPACT_STOP; }
// Behind a round's compaction every running trajectory sits in a slot below the count of running trajectories, and the last count
// the host has read is an upper bound of that (counts only fall): the kernels that follow are launched over that many slots instead
// of the whole batch (in its tail a batch of 65536 otherwise pays 53 us per k_linearize launch for 100 000 blocks that find nothing
// to do).  Whole groups of 64: k_linearize and k_rollout3 hand out 64 consecutive slots per wavefront.
inline long slots_in_use(long bound, unsigned seen_active) {
  const long want = std::max<long>(64, ((long)seen_active + 63) / 64 * 64);
  return std::min(bound, want);
}
int launch_compact(qilqr_solver *s, long B, long n) {
  launch(s, K_OTHER, k_compact_plan, dim3(1), dim3(1024), s->st, (int)B);
  const unsigned grid = std::min<unsigned>(cdiv(B, 2) * 2, 4096u);  // (work items: COMPACT_SPLIT per pair; the kernel strides over them)
  const int with_records = s->params.mu_init > 0.0 ? 1 : 0;  // a restart runs the recursion on the current records again
  if (s->f32)
    launch(s, K_OTHER, k_compact_move<float>, dim3(grid), dim3(256), s->st, (int)B, (int)n, s->compact_out, with_records);
  else
    launch(s, K_OTHER, k_compact_move<double>, dim3(grid), dim3(256), s->st, (int)B, (int)n, s->compact_out, with_records);
  return QILQR_OK;
}

// A batch of 1025 ... 4096 trajectories runs the same two kernels apart, with the compaction between them; once the running
// trajectories fit the combined launch -- `slots` of them for this (sub-)batch: a block of four per CU over all the sub-batches --
// the compaction has nothing left to give and the rounds change over to the one launch.
// Round 6: a batch BEYOND 4096 does the same from the round in which its rollouts are k_rollout16's anyway (launch_rollout: the 17th, or
// every round with single_wave_rollout = 3) -- the backward pass is one arithmetic in every form, so the combined launch's fused form gives
// the bits of the six-wavefront launches it replaces, and a problem's bits stay independent of its batch.
struct TailFuse {
  bool kinds = false;  // the round's kernels are the fused k_backward4 and k_rollout16 (or, from round `from` on, stand for the same bits)
  long slots = 0;      // slots in use at or below which this (sub-)batch's rounds are one launch
  unsigned stop = 0;   // the compaction runs while more trajectories than this are running
  long from = 0;       // first round in which the changeover may happen
};
#ifndef QILQR_LATE_TAIL
#define QILQR_LATE_TAIL 1  // (0: batches beyond 4096 keep three launches per round to the end -- A/B)
#endif
bool late_tail_kinds(const qilqr_solver *s, long B, long total_B, bool tiled, long *from) {
  const long load_B = std::max(B, total_B);
  if (!QILQR_LATE_TAIL || s->dev.round_launch == 1 || s->integrator != 0 || !s->symmetric || !tiled) return false;
  if (s->dev.force_general != 0 || load_B <= R16_MAX_B || backward_kind(s, load_B) != BW_FOUR) return false;
  if (s->dev.single_wave_rollout == 0) *from = ROLLOUT16_FROM;
  else if (s->dev.single_wave_rollout == 3) *from = 0;
  else return false;
  return true;
}
TailFuse tail_fuse(const qilqr_solver *s, long B, int nparts) {
  TailFuse t;
  t.stop = compact_stop(s);
  if (!s->compact || s->dev.compaction == 1) return t;  // (forced: the compaction runs to the last trajectory)
  t.kinds = fuse_kinds(s, B) || late_tail_kinds(s, B, s->total_B, s->st.layout.tiled != 0, &t.from);
  if (t.kinds) {
    // (a block of four per CU over the sub-batches of ONE stream, two per CU over two streams', three over three and more: measured once
    // the tail ran on k_round -- profiles/r06_ab.txt section 14: B = 4096 + 1 %, 8192 + 2-3 %, 16384 + 1.5 %; a single stream at two blocks
    // per CU loses 17 % at B = 2048, whose whole solve would then be the tail)
    t.slots = std::max<long>(64, std::min(nparts, 3) * 4L * s->num_cus / nparts / 64 * 64);
    t.stop = std::max<unsigned>(t.stop, (unsigned)t.slots);
  }
  return t;
}

int read_active(qilqr_solver *s, int *n_active) {
  HIP_TRY(hipMemcpyAsync(s->h_counters, s->st.counters + COUNT_BASE, sizeof(int) * COUNT_STRIPES, hipMemcpyDeviceToHost,
                         s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  int total = 0;
  for (int k = 0; k < COUNT_STRIPES; ++k) total += s->h_counters[k];
  *n_active = total;
  return QILQR_OK;
}

// ---- host-buffer batch solve: the copy-back under the tail of the solve.
// A batch takes as long as its slowest problem (configs[1]: 33 rounds for a mean of 12.5 iterations), and for the last third of
// the rounds nine trajectories in ten already have their exit status while the copy engines sit idle.  When the host sees
// the count of running trajectories fall to B / 8, it marks the finished ones between two rounds (k_mark_final, on the
// solver's stream), and a second stream gathers exactly those and copies the result arrays to the caller's (pinned) host
// buffers while the rounds of the others go on.  After the last round the late finishers -- at most B / 8 -- are gathered into
// a small compact block, copied, and put into their rows by the host.  The caller's arrays end up bit-identical to the
// one-piece copy.
int g_force_staged_late = 0;  // (diagnostics build: qilqr_debug_set_staged_late)
struct EarlyOut {
  double *h_traj, *h_cost;
  int32_t *h_status, *h_iters, *h_bwd, *h_fwd;
  // the same arrays as the DEVICE addresses them (pinned host memory is mapped: hipHostGetDevicePointer), when every one resolves: the late
  // finishers' rows are then written by k_gather straight into the caller's arrays over the link (round 6) -- no compact block, no second
  // copy, no scatter by the host
  bool direct = false;
  double *v_traj = nullptr, *v_cost = nullptr;
  int32_t *v_status = nullptr, *v_iters = nullptr, *v_bwd = nullptr, *v_fwd = nullptr;
  unsigned threshold = 0;  // fire when the active count is at or below this (and not zero)
  bool fired = false;
  int late_cap = 0;        // rows of the compact block: the active count seen when firing (the count only falls)
  struct LateLayout *layout = nullptr;
};
struct LateLayout {  // one block, device and host alike: [traj rows][cost][status | iters | n_bwd | n_fwd][idx][count]
  size_t traj, cost, ints, idx, count, bytes;
};
inline LateLayout late_layout(long rows, long n) {
  LateLayout L;
  L.traj = 0;
  L.cost = sizeof(double) * (size_t)rows * n * 18;
  L.ints = L.cost + sizeof(double) * rows;
  L.idx = L.ints + sizeof(int) * 4 * rows;
  L.count = L.idx + sizeof(int) * rows;
  L.bytes = L.count + sizeof(int) * 2;
  return L;
}
int ensure_early_buffers(qilqr_solver *s, long B, long n, long rows) {
  if (!s->early_stream) HIP_TRY(hipStreamCreateWithFlags(&s->early_stream, hipStreamNonBlocking));
  if (!s->early_evt) HIP_TRY(hipEventCreateWithFlags(&s->early_evt, hipEventDisableTiming));
  if (!s->early_done) HIP_TRY(hipEventCreateWithFlags(&s->early_done, hipEventDisableTiming));
  if ((size_t)B > s->early_cap) {
    if (s->d_early) (void)hipFree(s->d_early);
    s->d_early = nullptr;
    s->early_cap = 0;
    HIP_TRY(hipMalloc((void **)&s->d_early, sizeof(int) * 2 * (size_t)B));
    s->early_cap = (size_t)B;
  }
  const size_t want = late_layout(rows, n).bytes;
  if (want > s->late_bytes) {
    if (s->d_late) (void)hipFree(s->d_late);
    if (s->h_late) (void)hipHostFree(s->h_late);
    s->d_late = s->h_late = nullptr;
    s->late_bytes = 0;
    HIP_TRY(hipMalloc((void **)&s->d_late, want));
    HIP_TRY(hipHostMalloc((void **)&s->h_late, want, hipHostMallocDefault));
    s->late_bytes = want;
  }
  return QILQR_OK;
}
int gather(qilqr_solver *s, long B, long n, double *d_traj, double *d_cost, int *d_status, int *d_iters, int *d_bwd, int *d_fwd,
           const int *mask, int want, const int *row_of);
// called from the polling loop of run_solve when the count has fallen to the threshold
int fire_early_out(qilqr_solver *s, long B, long n, EarlyOut *eo, unsigned active) {
  eo->fired = true;
  eo->late_cap = (int)active;
  *eo->layout = late_layout((long)active, n);
  int *early = s->d_early, *late_count = (int *)(s->d_late + eo->layout->count);
  launch(s, K_OTHER, k_mark_final, dim3(cdiv(B, 256)), dim3(256), s->st, (int)B, early, late_count);
  HIP_TRY(hipEventRecord(s->early_evt, s->stream));
  HIP_TRY(hipStreamWaitEvent(s->early_stream, s->early_evt, 0));
  hipStream_t main_stream = s->stream;
  s->stream = s->early_stream;  // (launch() goes to s->stream)
  int *d_int = s->stage_int;
  const int rc = gather(s, B, n, eo->h_traj ? s->stage_traj : nullptr, eo->h_cost ? s->stage_cost : nullptr, eo->h_status ? d_int : nullptr,
                        eo->h_iters ? d_int + B : nullptr, eo->h_bwd ? d_int + 2 * B : nullptr, eo->h_fwd ? d_int + 3 * B : nullptr, early, 1,
                        nullptr);
  s->stream = main_stream;
  if (rc) return rc;
  hipStream_t es = s->early_stream;
  if (eo->h_traj) HIP_TRY(hipMemcpyAsync(eo->h_traj, s->stage_traj, sizeof(double) * 18 * (size_t)B * n, hipMemcpyDeviceToHost, es));
  if (eo->h_cost) HIP_TRY(hipMemcpyAsync(eo->h_cost, s->stage_cost, sizeof(double) * B, hipMemcpyDeviceToHost, es));
  if (eo->h_status) HIP_TRY(hipMemcpyAsync(eo->h_status, d_int, sizeof(int) * B, hipMemcpyDeviceToHost, es));
  if (eo->h_iters) HIP_TRY(hipMemcpyAsync(eo->h_iters, d_int + B, sizeof(int) * B, hipMemcpyDeviceToHost, es));
  if (eo->h_bwd) HIP_TRY(hipMemcpyAsync(eo->h_bwd, d_int + 2 * B, sizeof(int) * B, hipMemcpyDeviceToHost, es));
  if (eo->h_fwd) HIP_TRY(hipMemcpyAsync(eo->h_fwd, d_int + 3 * B, sizeof(int) * B, hipMemcpyDeviceToHost, es));
  HIP_TRY(hipEventRecord(s->early_done, es));
  return QILQR_OK;
}

// The outer loop of ILQR::solve (ilqr.hh:53-87) for trajectories already in st.traj[0].
// on_round is called behind every round's backward pass (and rollout), in both modes: debug capture enqueues its kernel there.
// drain = false: return as soon as the host knows that no trajectory is active; the caller enqueues
// its own work behind the rounds still in flight and waits for the stream itself.
// on_count (optional): called with the count of running trajectories each time the free-running loop learns one.
// double_ok: the caller does not look at the solve round by round (on_round does nothing): launches may hold two rounds
template <typename F>
int run_solve(qilqr_solver *s, long B, long n, int sync_every, F on_round, bool drain = true,
              const std::function<int(unsigned)> *on_count = nullptr, bool double_ok = false) {
  int rc;
  s->round_captured = false;
  if ((rc = launch_linearize(s, B, n, 0, 0))) return rc;
  {
    launch(s, K_OTHER, k_init, dim3(cdiv(B, 64)), dim3(64), s->params, s->st, (int)B, (int)n);
  }
  if (!(0.0 < s->params.max_iters)) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    return QILQR_OK;
  }
  // a trajectory needs at most max_iters backward passes and max_iters * ls_max_iters trials
  const double bound = (std::fmin(s->params.max_iters, 1e7) + 1.0) * ((double)std::max(s->params.ls_max_iters, 1) + 1.0) * (1.0 + max_restarts(s->params));
  const long max_rounds = (long)std::fmin(bound, 2e9);
  const int lag = (sync_every > 1) ? std::min(sync_every, 6) : 0;
  if (lag == 0) {
    // Synchronous rounds (debug capture): the host reads the count after every k_backward.
    for (long round = 0; round < max_rounds; ++round) {
      // k_backward first settles the candidate of the previous round (cost, Armijo, convergence) and
      // counts the trajectories still active; then rollout + linearise the next candidates
      if ((rc = launch_backward(s, B, n, 0))) return rc;
      int active = 0;
      if ((rc = read_active(s, &active))) return rc;
      if ((rc = on_round())) return rc;
      if (active == 0) break;
      if ((rc = launch_rollout(s, B, n, F_SEARCH, round))) return rc;
      if ((rc = launch_linearize(s, B, n, 1, F_SEARCH))) return rc;
    }
  } else {
    // Free-running rounds: three kernels per round and nothing else on the stream.  k_linearize hands
    // the count of still-active trajectories to the host through pinned memory, tagged with its round;
    // the host looks at the count `lag` rounds late, i.e. keeps the stream `lag` rounds ahead of the
    // device, so the GPU never waits for a host round trip.  Rounds enqueued past the end find nothing
    // to do.
    // the counter set of a round's parity (k_round publishes a round's count from the NEXT launch; the other kernels of a round
    // count and publish within it, in the same set)
    struct CounterSet {
      qilqr_solver *s;
      int *base;
      CounterSet(qilqr_solver *s_, long round, bool two) : s(s_), base(s_->st.counters) { if (two) s->st.counters = base + (round & 1) * COUNT_WORDS; }
      ~CounterSet() { s->st.counters = base; }
    };
    for (int k = 0; k < 8; ++k) s->h_active[k] = 0;
    const InFlight in_flight(s->device);
    const bool can_fuse = fuse_backward_rollout(s, B) && !s->compact;  // (compaction works between the two halves)
    const TailFuse tf = tail_fuse(s, B, 1);
    if (s->compact) s->plan_heads.push_back(0);
    unsigned seen_active = (unsigned)B;  // the last count the host has read (the count only falls)
    long used = B;                       // slots the round's kernels are launched over (slots_in_use)
    bool pending_publish = false;        // the round before was a k_round: the next launch publishes its count
    unsigned launched_rounds[8] = {1, 1, 1, 1, 1, 1, 1, 1};  // rounds in launch `round & 7` (its count is the sum of theirs)
    bool two_sets = false;               // a k_round has run in this solve: rounds count into the counter set of their parity
    bool tail_started = false;           // a compacted batch has changed over to the combined launch for the rest of the solve
    for (long round = 0; round < max_rounds; ++round) {
      const RoctxRange range(s, "round", round);
      // (one more compaction behind the last count above the threshold brings the slots in use under it)
      // (once the compaction has stopped for a batch that changes over to the combined launch it stays stopped: the launches may then hold
      // several rounds, and the sums of counts they report say nothing against the threshold)
      const bool compacting = s->compact && !tail_started && (seen_active > tf.stop || (tf.kinds && used > tf.slots));
      if (s->compact && tf.kinds && !compacting && used <= tf.slots && round >= tf.from) tail_started = true;
      s->live_hint = (long)seen_active;
      const bool fuse_now = (can_fuse || tail_started) && in_flight.alone(s->dev.fuse_in_flight == 1);
      // (until round 6 k_round linearised a block's candidates AFTER the rollout, 2.5 times slower than k_linearize with four candidates per
      // block and idle CUs beside it, and B = 64 ... 512 took k_backward_rollout + k_linearize in their first rounds; with the linearisation
      // behind the rollout -- round_follow -- k_round is ahead at every size: B = 64 + 2.1 %, 128 + 2.3 %, 256 + 3 %, 512 + 3.6 %)
      if (fuse_now && round_kernel_ok(s)) {
        // several rounds per launch where the rounds are this kernel for the rest of the solve (no compaction any more, whose
        // thresholds go by the count) and the caller does not look at a solve round by round (the single solve's debug capture)
        const int rounds = ((can_fuse || tail_started) && double_ok) ? rounds_per_launch(s) : 1;
        // the backward pass's form by how many trajectories a block holds on average (the same bits: round_kernels.h)
        // (and only in launches of several rounds: with one round per launch the 384-thread form measured slower)
        const bool six = s->dev.force_general == 8 || (s->dev.force_general == 0 && rounds > 1 && 2L * (long)seen_active <= cdiv(used, 4) * 4L);
        if ((rc = launch_round(s, used, n, round, pending_publish, rounds, six))) return rc;
        s->round_captured = true;
        launched_rounds[round & 7] = rounds;
        pending_publish = true;
        two_sets = true;
        if ((rc = on_round())) return rc;
        goto round_enqueued;
      }
      if (pending_publish) {  // the round before was a k_round: its count has no launch left to publish it
        launch(s, K_OTHER, k_publish_active, dim3(1), dim3(64), s->st.counters + ((round + 1) & 1) * COUNT_WORDS, s->st.host_active,
               (int)((round - 1) & 0x3fffffff));
        pending_publish = false;
      }
      launched_rounds[round & 7] = 1;
      s->round_captured = false;
      {
      const CounterSet counter_set(s, round, two_sets);
      if (fuse_now) {
        if ((rc = launch_backward_rollout(s, used, n))) return rc;
      } else {
        if ((rc = launch_backward(s, used, n, 0))) return rc;
        if (compacting) {
          if ((rc = launch_compact(s, used, n))) return rc;
          used = slots_in_use(used, seen_active);
        }
        if ((rc = launch_rollout(s, used, n, F_SEARCH, round))) return rc;
      }
      if ((rc = on_round())) return rc;  // (debug capture of the single solve: one more launch, nothing waited for)
      if ((rc = launch_linearize(s, used, n, 1, F_SEARCH, (int)(round & 0x3fffffff)))) return rc;
      }
    round_enqueued:
      // (following launches of several rounds ONE launch back instead of two -- four rounds that find nothing to do at the end of a solve
      // instead of eight -- measured no different: 4.71-4.73 ms either way)
      if (round >= lag) {
        const long old = round - lag;
        const unsigned tag = (unsigned)((old & 0x3fffffff) + 1);
        unsigned long long v;
        for (long spins = 0;; ++spins) {
          v = __atomic_load_n(&s->h_active[old & 7], __ATOMIC_ACQUIRE);
          if ((unsigned)(v >> 32) == tag) break;
          if ((spins & 1023) == 1023) {
            // never spin on a dead stream: a drained or failed stream without the tag is an error
            const hipError_t q = hipStreamQuery(s->stream);
            if (q != hipErrorNotReady) {
              v = __atomic_load_n(&s->h_active[old & 7], __ATOMIC_ACQUIRE);
              if ((unsigned)(v >> 32) == tag) break;
              return fail(QILQR_ERR_HIP, std::string("a round never reported its active count: ") + hipGetErrorString(q));
            }
          }
          __builtin_ia32_pause();
        }
        if ((unsigned)v == 0) break;
        // a launch of several rounds reports the SUM of their counts; counts only fall, so the mean over the launch's rounds is an upper
        // bound of the last round's -- of the count now
        const unsigned now_at_most = ((unsigned)v + launched_rounds[old & 7] - 1) / launched_rounds[old & 7];
        seen_active = now_at_most;
        // a block that gave up a hand-off (BatchState::host_error) voids the call: stop enqueuing rounds on void gains -- each
        // could burn a full bounded spin -- let what is in flight finish, and report
        if (__atomic_load_n(s->h_active + 8 * (1 + qilqr_solver::MAX_PARTS), __ATOMIC_ACQUIRE)) {
          (void)hipStreamSynchronize(s->stream);
          if (s->early_stream) (void)hipStreamSynchronize(s->early_stream);
          return device_error(s);
        }
        if (on_count && (rc = (*on_count)(now_at_most))) return rc;
        // (Round 3 tried following the device ONE round behind in the tail, where a round takes well over 100 us and the host
        // needs about 15 to enqueue the next: one round of three empty launches fewer after the last trajectory has finished --
        // 36 rounds instead of 37 -- and no measurable difference, 5.223 against 5.221 ms per solve.  `lag` stays fixed.)
      }
    }
  }
  if (drain) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    return device_error(s);
  }
  return QILQR_OK;
}

// ---- sub-batches on their own streams
// The three kernels of a round are bound by three different things (serial latency and the matrix pipe,
// serial latency at a handful of wavefronts, HBM writes), and one stream runs them one after the other.
// A batch is therefore cut into `parts` contiguous ranges of 64-trajectory tiles, each with its own
// stream, counters and host hand-off words, whose rounds run independently: while one part is in its
// rollout another is in its backward pass and a third writes its knot records.  Trajectories are
// independent, so the results are those of the single-stream solve.
struct Part {
  hipStream_t stream;
  BatchState st;  // the solver's workspace seen from the part's first trajectory
  long nb;        // trajectories in the part
  unsigned long long *h_active;
  bool done;
  unsigned seen_active;  // the last count of running trajectories the host has read
  long used;             // slots its kernels are launched over (slots_in_use)
  bool tail = false;          // the part has changed over to the combined launch for the rest of the solve (tail_fuse)
  bool round_kernel = false;  // ... and from there to k_round, several rounds per launch
  unsigned launched_rounds[8] = {1, 1, 1, 1, 1, 1, 1, 1};  // rounds in launch `round & 7` (its count is the sum of theirs)
};
// the workspace of trajectories [b0, b0 + nb), b0 a multiple of 64
BatchState slice_state(const qilqr_solver *s, long b0, long n, int part) {
  const BatchState &w = s->st;
  BatchState v = w;
  const size_t es = s->f32 ? sizeof(float) : sizeof(double);
  auto adv = [&](const void *p, long elems) { return (void *)((char *)p + (size_t)elems * es); };
  for (int k = 0; k < 2; ++k) {
    v.traj[k] = adv(w.traj[k], knot_base<true>(b0, n, 18));
    v.lin[k] = adv(w.lin[k], rec_base(w.layout, b0, n));  // (b0 is a multiple of 64: the same offset in either placement)
    v.knot_cost[k] = w.knot_cost[k] + cost_index(b0, 0, n);
  }
  v.gains = adv(w.gains, knot_base<true>(b0, n, 52));
  if (w.desired_tiled) v.desired = adv(w.desired, knot_base<true>(b0, n, 18));
  v.cur = w.cur + b0; v.cost = w.cost + b0; v.prev_cost = w.prev_cost + b0; v.terms = w.terms + 2 * b0;
  v.alpha = w.alpha + b0; v.mu = w.mu + b0; v.trial = w.trial + b0; v.flags = w.flags + b0; v.status = w.status + b0;
  v.iters = w.iters + b0; v.n_bwd = w.n_bwd + b0; v.n_fwd = w.n_fwd + b0;
  v.counters = s->d_part_counters + 2 * COUNT_WORDS * part;  // (two sets: k_init zeroes both, k_round alternates)
  v.host_active = s->d_active + 8 * (1 + part);
  if (w.cost_hist) v.cost_hist = w.cost_hist + b0 * w.hist_cap;
  v.dump = adv(w.dump, 4 * b0);
  v.orig = w.orig + b0;
  v.row0 = w.row0 + (int)b0;
  v.plan = w.plan + PLAN_HEAD * (part + 1) + 4 * b0;  // (part p's plan ends where part p + 1's begins)
  if (w.stamps) v.stamps = w.stamps + 8 * b0;
  return v;
}
// launch_* work on s->st / s->stream: point them at a part for the duration of a scope
struct PartScope {
  qilqr_solver *s;
  BatchState st0;
  hipStream_t stream0;
  PartScope(qilqr_solver *s_, const Part &p) : s(s_), st0(s_->st), stream0(s_->stream) {
    s->st = p.st;
    s->stream = p.stream;
  }
  ~PartScope() {
    s->st = st0;
    s->stream = stream0;
  }
};
// hardware queues HIP multiplexes this process's streams onto: GPU_MAX_HW_QUEUES as the runtime read it at start-up (default 4)
// Latched at the first qilqr_create_sized of the process, which calls it (the runtime reads the variable once, when it starts: a value
// put into the environment later -- os.environ after the first GPU call -- changes nothing in the runtime and must change nothing here)
int hw_queues() {
  static const int latched = [] {
    const char *e = std::getenv("GPU_MAX_HW_QUEUES");
    const int q = e ? std::atoi(e) : 4;
    return q > 0 ? q : 4;
  }();
  return latched;
}
int auto_parts(const qilqr_solver *s, long B) {
  const long tiles = (B + 63) / 64;
  // Measured (MI355X, N = 100): up to a few thousand trajectories every kernel is latency-bound and sharing
  // SIMDs with another part's kernels only slows both (B = 1024, round 3: 194k solves/s on one stream, 168k on two,
  // 153k on four); from 4096 on two parts gain 4-5%.  Between 4096 and 16384 FOUR parts are better still when
  // every part's stream has a hardware queue of its own -- GPU_MAX_HW_QUEUES=8 in the environment before the runtime
  // starts (bench.py sets it): 5120: 366k against 356k solves/s, 6144: 403k / 377k, 7168: 437k / 406k,
  // 8192: 461k / 429k, 10240: 441k / 424k, 12288: 472k / 460k; level at 4096, 16384 and 65536; with HIP's default four queues
  // the parts collide with each other and with the caller's streams and two are the safer choice.
  // Round 4 (compaction, k_backward4 and k_rollout3 at every size beyond 4096): four parts are ahead at 16384 and 65536 as well
  // (596k against 591k, 698k against 686k).
  int want = s->dev.streams > 0 ? s->dev.streams : (B >= REGIME_B ? ((B > REGIME_B && hw_queues() >= 8) ? 4 : 2) : 1);
  if (want > qilqr_solver::MAX_PARTS) want = qilqr_solver::MAX_PARTS;
  while (want > 1 && tiles < 2 * want) --want;  // at least two tiles per part
  return want;
}
// streams and completion events of the first nparts sub-batches, created on first use
int ensure_parts(qilqr_solver *s, int nparts) {
  for (int k = 0; k < nparts; ++k) {
    if (!s->part_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&s->part_stream[k], hipStreamNonBlocking));
    if (!s->part_done[k]) HIP_TRY(hipEventCreateWithFlags(&s->part_done[k], hipEventDisableTiming));
  }
  return QILQR_OK;
}
// The outer loop of ILQR::solve for a batch cut into parts (free-running rounds only).  On return the
// main stream waits for every part; the caller enqueues its own work there.
int run_solve_parts(qilqr_solver *s, long B, long n, int nparts) {
  int rc;
  if ((rc = ensure_parts(s, nparts))) return rc;
  const long tiles = (B + 63) / 64;
  std::vector<Part> parts;
  for (int p = 0; p < nparts; ++p) {
    const long t0 = tiles * p / nparts, t1 = tiles * (p + 1) / nparts;
    const long b0 = t0 * 64, b1 = std::min(t1 * 64, B);
    Part part;
    part.stream = s->part_stream[p];
    part.st = slice_state(s, b0, n, p);
    part.nb = b1 - b0;
    part.h_active = s->h_active + 8 * (1 + p);
    part.done = false;
    part.seen_active = (unsigned)part.nb;
    part.used = part.nb;
    if (s->compact) s->plan_heads.push_back(part.st.plan - s->st.plan);
    for (int k = 0; k < 8; ++k) part.h_active[k] = 0;
    parts.push_back(part);
  }
  // the parts start when the main stream has tiled the inputs
  HIP_TRY(hipEventRecord(s->main_ready, s->stream));
  for (auto &part : parts) HIP_TRY(hipStreamWaitEvent(part.stream, s->main_ready, 0));
  for (auto &part : parts) {
    PartScope scope(s, part);
    if ((rc = launch_linearize(s, part.nb, n, 0, 0))) return rc;
    launch(s, K_OTHER, k_init, dim3(cdiv(part.nb, 64)), dim3(64), s->params, s->st, (int)part.nb, (int)n);
  }
  int remaining = nparts;
  if (0.0 < s->params.max_iters) {
    const double bound =
        (std::fmin(s->params.max_iters, 1e7) + 1.0) * ((double)std::max(s->params.ls_max_iters, 1) + 1.0) * (1.0 + max_restarts(s->params));
    const long max_rounds = (long)std::fmin(bound, 2e9);
    const int lag = std::max(1, std::min(s->dev.sync_every, 6));
    const InFlight in_flight(s->device);
    const TailFuse tf = tail_fuse(s, B, nparts);
    for (long round = 0; round < max_rounds && remaining > 0; ++round) {
      long live_all = 0;  // (every part's last count: what shares the chip with this part's kernels)
      for (auto &part : parts) live_all += part.done ? 0 : (long)part.seen_active;
      s->live_hint = std::max<long>(live_all, 1);
      for (auto &part : parts) {
        if (part.done) continue;
        PartScope scope(s, part);
        const RoctxRange range(s, "round", round, (long)(&part - &parts[0]));
        // (compacting in EVERY round while it runs is right: waiting until 1/16, 1/8 or 1/4 of the slots in use are known holes measured -2 / -4 /
        // -5 % at B = 8192 and -9 / -9 / -12 % at 65536 -- profiles/r06_ab.txt)
        const bool compacting = s->compact && !part.tail && (part.seen_active > tf.stop || (tf.kinds && part.used > tf.slots));
        if (tf.kinds && !compacting && part.used <= tf.slots && round >= tf.from) part.tail = true;  // (counts and slots only fall)
        part.launched_rounds[round & 7] = 1;
        if (part.tail && (part.round_kernel || in_flight.alone(s->dev.fuse_in_flight == 1))) {
          // as in run_solve: k_round, four rounds per launch (fp64; the mixed mode keeps the combined launch and k_linearize).  One way only,
          // so the round before the first k_round has been published by its own k_linearize and every later one by the k_round behind it.
          if (round_kernel_ok(s) && QILQR_LATE_TAIL) {
            const int rounds = rounds_per_launch(s);
            const bool six = rounds > 1 && 2L * (long)part.seen_active <= cdiv(part.used, 4) * 4L;
            if ((rc = launch_round(s, part.used, n, round, part.round_kernel, rounds, six))) return rc;
            part.launched_rounds[round & 7] = (unsigned)rounds;
            part.round_kernel = true;
            continue;
          }
          if ((rc = launch_backward_rollout(s, part.used, n))) return rc;
          if ((rc = launch_linearize(s, part.used, n, 1, F_SEARCH, (int)(round & 0x3fffffff)))) return rc;
          continue;
        }
        if ((rc = launch_backward(s, part.used, n, 0))) return rc;
        if (compacting) {
          if ((rc = launch_compact(s, part.used, n))) return rc;
          part.used = slots_in_use(part.used, part.seen_active);
        }
        if ((rc = launch_rollout(s, part.used, n, F_SEARCH, round))) return rc;
        if ((rc = launch_linearize(s, part.used, n, 1, F_SEARCH, (int)(round & 0x3fffffff)))) return rc;
      }
      if (round < lag) continue;
      const long old = round - lag;
      const unsigned tag = (unsigned)((old & 0x3fffffff) + 1);
      for (auto &part : parts) {
        if (part.done) continue;
        unsigned long long v;
        for (long spins = 0;; ++spins) {
          v = __atomic_load_n(&part.h_active[old & 7], __ATOMIC_ACQUIRE);
          if ((unsigned)(v >> 32) == tag) break;
          if ((spins & 1023) == 1023) {
            const hipError_t q = hipStreamQuery(part.stream);
            if (q != hipErrorNotReady) {
              v = __atomic_load_n(&part.h_active[old & 7], __ATOMIC_ACQUIRE);
              if ((unsigned)(v >> 32) == tag) break;
              return fail(QILQR_ERR_HIP, std::string("a round never reported its active count: ") + hipGetErrorString(q));
            }
          }
          __builtin_ia32_pause();
        }
        // (a launch of several rounds reports the sum of their counts: the mean bounds the last round's)
        part.seen_active = ((unsigned)v + part.launched_rounds[old & 7] - 1) / part.launched_rounds[old & 7];
        if ((unsigned)v == 0) {
          part.done = true;
          --remaining;
        }
      }
      // a block that gave up a hand-off (BatchState::host_error) voids the call: stop enqueuing rounds on void gains -- each could burn a
      // full bounded spin in every block that gave up -- let what is in flight finish, and report (as run_solve does)
      if (__atomic_load_n(s->h_active + 8 * (1 + qilqr_solver::MAX_PARTS), __ATOMIC_ACQUIRE)) {
        for (auto &part : parts) (void)hipStreamSynchronize(part.stream);
        return device_error(s);
      }
    }
  }
  for (int p = 0; p < nparts; ++p) {
    HIP_TRY(hipEventRecord(s->part_done[p], parts[p].stream));
    HIP_TRY(hipStreamWaitEvent(s->stream, s->part_done[p], 0));
  }
  return QILQR_OK;
}

int gather(qilqr_solver *s, long B, long n, double *d_traj, double *d_cost, int *d_status, int *d_iters,
           int *d_bwd, int *d_fwd, const int *mask = nullptr, int want = 0, const int *row_of = nullptr) {
  if (s->f32)
    launch(s, K_OTHER, k_gather<float>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * 9 * TILE, 256)), dim3(256), s->st, (int)B, (int)n,
                       d_traj, d_cost, d_status, d_iters, d_bwd, d_fwd, mask, want, row_of);
  else
    launch(s, K_OTHER, k_gather<double>, dim3((unsigned)cdiv(B, TILE), (unsigned)cdiv(n * 9 * TILE, 256)), dim3(256), s->st, (int)B, (int)n,
                       d_traj, d_cost, d_status, d_iters, d_bwd, d_fwd, mask, want, row_of);
  return QILQR_OK;
}

int check_quaternions(const double *traj, long count, const char *what) {
  // manif's SO3 constructor rejects quaternions that are not unit within 1e-10
  for (long i = 0; i < count; ++i) {
    const double *q = traj + i * 18 + 4;
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(std::fabs(nn - 1.0) <= 1e-10))
      return fail(QILQR_ERR_BAD_QUATERNION, std::string(what) + ": quaternion not normalized at knot " + std::to_string(i));
  }
  return QILQR_OK;
}

// The persistent solve (solve4.h): every trajectory from its first linearisation to its exit status in ONE launch.
// Requirements: symmetric weights (the matrix-core recursion of k_backward4), no per-round host visibility (debug capture
// of trajectories uses the rounds).  qilqr_device_config.persistent: 0 = by measurement, 1 = always, 2 = never.
// By measurement (profiles/microbench/persistent_sweep.py, MI355X, N = 100, device-resident, ms per batch solve) the rounds are
// level or ahead at every batch size but one -- 256: 4.15 vs 4.45, 1024: 5.63 vs 5.74, 1536: 7.52 vs 7.21, 2048: 8.1 vs 9.5,
// 8192: 22.0 vs 27.6: both paths are bound by (iterations of the slowest trajectory) x (latency of one iteration), and inside
// k_solve4 the forward phase ends one linearisation task (~11 us) after the rollout while its step waves run beside three
// linearising wavefronts.
// So 0 selects the rounds; the persistent solve stays selectable and tested.
bool use_persistent(const qilqr_solver *s, long B) {
  (void)B;
#ifdef QILQR_WITH_SOLVE4
  return s->symmetric && s->dev.persistent == 1 && s->integrator == 0;
#else
  (void)s;
  return false;  // k_solve4 is in the diagnostics build (qilqr_create refuses persistent = 1 here)
#endif
}
#ifdef QILQR_WITH_SOLVE4
int launch_solve4(qilqr_solver *s, long B, long n) {
  const unsigned groups = cdiv(B, 4);
  const unsigned grid = std::min<unsigned>(groups, (unsigned)s->num_cus);  // one block per CU (256 VGPRs, 105 KB of LDS); the rest queue
#define QILQR_LAUNCH_S4(S, LK) \
  launch(s, K_SOLVE, k_solve4<S, LK>, dim3(grid), dim3(S4_THREADS), s->consts, (const ModelConsts<S> *)s->d_consts, s->params, s->st, (int)B, (int)n, 0u)
  switch (layout_kind(s->layout) + (s->f32 ? 3 : 0)) {
    case 0: QILQR_LAUNCH_S4(double, 0); break;
    case 1: QILQR_LAUNCH_S4(double, 1); break;
    case 2: QILQR_LAUNCH_S4(double, 2); break;
    case 3: QILQR_LAUNCH_S4(float, 0); break;
    case 4: QILQR_LAUNCH_S4(float, 1); break;
    default: QILQR_LAUNCH_S4(float, 2); break;
  }
#undef QILQR_LAUNCH_S4
  return QILQR_OK;
}
#else
int launch_solve4(qilqr_solver *, long, long) { return fail(QILQR_ERR_INVALID_ARG, "k_solve4 is in the diagnostics build"); }
#endif

// The batch solve on device-resident buffers.  drain = false: return with the gather enqueued, the caller puts
// its own copies behind it and waits for the stream itself.
int solve_batch_device_impl(qilqr_solver *s, const double *d_init, const double *d_desired_batch, int32_t B, int32_t n,
                            double *d_out_traj, double *d_out_cost, int32_t *d_out_status, int32_t *d_out_iters,
                            int32_t *d_out_n_bwd, int32_t *d_out_n_fwd, bool drain) {
  if (!s || !d_init) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  const RoctxRange range(s, "batch solve, trajectories:", (long)B);
  int rc = begin_batch(s, B, n, d_desired_batch);
  if (rc) return rc;
  const bool persistent = use_persistent(s, B);
  if ((rc = to_tiled(s, d_init, s->st.traj[0], B, n, 18, persistent ? s->st.counters : nullptr))) return rc;
  const int nparts = (s->dev.sync_every > 1) ? auto_parts(s, B) : 1;
  // compaction: free-running rounds only (the host never waits for a plan), not beside the copy-back under the tail (it gathers by
  // slot), the per-iteration cost history (rows by slot) or per-problem desired trajectories (they would have to move along)
  s->compact = s->dev.compaction >= 0 && s->dev.sync_every > 1 && !persistent && !s->st.cost_hist && !s->st.desired_tiled && !s->early_out &&
               0.0 < s->params.max_iters &&
               (s->dev.compaction == 1 || (!fuse_backward_rollout(s, B) && backward_kind(s, B) != BW_ONE && s->st.layout.tiled));
  s->compact_out = CompactOut{d_out_traj, d_out_cost, d_out_status, d_out_iters, d_out_n_bwd, d_out_n_fwd};
  s->plan_heads.clear();
  struct CompactScope {  // (every return below leaves the flag off for the other entry points)
    qilqr_solver *s;
    ~CompactScope() { s->compact = false; }
  } compact_scope{s};
  if (persistent) {
    if ((rc = launch_solve4(s, B, n))) return rc;
  } else if (nparts > 1) {
    if ((rc = run_solve_parts(s, B, n, nparts))) return rc;
  } else {
    EarlyOut *eo = static_cast<EarlyOut *>(s->early_out);
    const std::function<int(unsigned)> hook = [&](unsigned active) -> int {
      if (eo->fired || active > eo->threshold) return QILQR_OK;
      return fire_early_out(s, B, n, eo, active);
    };
    if ((rc = run_solve(s, B, n, s->dev.sync_every, [] { return QILQR_OK; }, false, eo ? &hook : nullptr, /*double_ok=*/true))) return rc;
    if (eo && eo->fired && eo->direct) {
      // the late finishers' rows straight into the caller's (mapped, pinned) arrays, behind the early part's copies -- which cover every
      // row of those arrays, the late ones with stale data -- so that nothing overwrites them afterwards
      HIP_TRY(hipStreamWaitEvent(s->stream, s->early_done, 0));
      return gather(s, B, n, eo->v_traj, eo->v_cost, eo->v_status, eo->v_iters, eo->v_bwd, eo->v_fwd, s->d_early, 0, nullptr);
    }
    if (eo && eo->fired) {
      // the late finishers into the compact block, one copy to the pinned host block; qilqr_solve_batch puts them in place
      const LateLayout L = *eo->layout;
      int *early = s->d_early, *late_slot = early + s->early_cap, *late_idx = (int *)(s->d_late + L.idx), *late_count = (int *)(s->d_late + L.count);
      launch(s, K_OTHER, k_late_slots, dim3(cdiv(B, 256)), dim3(256), (int)B, (const int *)early, late_count, late_idx, late_slot, eo->late_cap);
      int *li = (int *)(s->d_late + L.ints);
      const long R = eo->late_cap;
      if ((rc = gather(s, B, n, eo->h_traj ? (double *)(s->d_late + L.traj) : nullptr, eo->h_cost ? (double *)(s->d_late + L.cost) : nullptr,
                       eo->h_status ? li : nullptr, eo->h_iters ? li + R : nullptr, eo->h_bwd ? li + 2 * R : nullptr,
                       eo->h_fwd ? li + 3 * R : nullptr, early, 0, late_slot)))
        return rc;
      HIP_TRY(hipMemcpyAsync(s->h_late, s->d_late, L.bytes, hipMemcpyDeviceToHost, s->stream));
      return QILQR_OK;  // (drain is false on this path: the caller waits for both streams)
    }
  }
  if ((rc = gather(s, B, n, d_out_traj, d_out_cost, d_out_status, d_out_iters, d_out_n_bwd, d_out_n_fwd, nullptr, 0,
                   s->compact ? s->st.orig : nullptr)))  // (with compaction: by the row a slot's trajectory came from)
    return rc;
  if (!drain) return QILQR_OK;
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipGetLastError());
  if (s->dev.profile) drain_events(s);
  return device_error(s);
}

// The host-buffer batch solve up to, but not including, the copies back: checks, staging buffers (kept between calls, no
// hipMalloc / hipFree per call), H2D, the solve, the gather into s->stage_traj / stage_cost / stage_int -- everything
// enqueued on the solver's stream, nothing waited for.  The copies are plain hipMemcpyAsync: direct DMA when the caller's
// buffers are pinned (qilqr_host_alloc, or any hipHostMalloc / hipHostRegister'ed memory), HIP's own chunked staging when
// they are pageable.
int solve_batch_staged(qilqr_solver *s, const double *init, const double *desired_batch, int32_t B, int32_t n) {
  if (!s || !init) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || n <= 0) return fail(QILQR_ERR_INVALID_ARG, "B and n must be positive");
  if (!desired_batch && n > s->n_desired)
    return fail(QILQR_ERR_LENGTH_MISMATCH, "trajectory longer than desired trajectory");
  HIP_TRY(hipSetDevice(s->device));
  const size_t cnt = 18 * (size_t)B * n, tb = sizeof(double) * cnt;
  auto grow = [&](auto **p, size_t *cap, size_t want, size_t elem) -> hipError_t {
    if (want <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)p, want * elem);
    if (e == hipSuccess) *cap = want;
    return e;
  };
  hipError_t e = grow(&s->stage_traj, &s->stage_traj_cap, cnt, sizeof(double));
  if (e == hipSuccess && desired_batch) e = grow(&s->stage_des, &s->stage_des_cap, cnt, sizeof(double));
  if (e == hipSuccess && (size_t)B > s->stage_B_cap) {
    if (s->stage_cost) (void)hipFree(s->stage_cost);
    if (s->stage_int) (void)hipFree(s->stage_int);
    s->stage_cost = nullptr;
    s->stage_int = nullptr;
    s->stage_B_cap = 0;
    e = hipMalloc((void **)&s->stage_cost, sizeof(double) * B);
    if (e == hipSuccess) e = hipMalloc((void **)&s->stage_int, sizeof(int) * 4 * B);
    if (e == hipSuccess) s->stage_B_cap = B;
  }
  // The uploads are enqueued first and the quaternion checks (manif's constructor check, SURVEY.md 8b: 0.1-0.2 ms of host
  // time for 1024 x 100 knots) run while the copy engine works; nothing that computes is enqueued before they have passed.
  if (e == hipSuccess) e = hipMemcpyAsync(s->stage_traj, init, tb, hipMemcpyHostToDevice, s->stream);
  if (e == hipSuccess && desired_batch) e = hipMemcpyAsync(s->stage_des, desired_batch, tb, hipMemcpyHostToDevice, s->stream);
  if (e != hipSuccess) return fail(QILQR_ERR_HIP, std::string("staging: ") + hipGetErrorString(e));
  int rc;
  if ((rc = check_quaternions(init, (long)B * n, "initial trajectory")) ||
      (desired_batch && (rc = check_quaternions(desired_batch, (long)B * n, "desired trajectory")))) {
    (void)hipStreamSynchronize(s->stream);  // the uploads read the caller's buffers: finished before the error returns
    return rc;
  }
  int *d_int = s->stage_int;
  return solve_batch_device_impl(s, s->stage_traj, desired_batch ? s->stage_des : nullptr, B, n, s->stage_traj, s->stage_cost, d_int,
                                 d_int + B, d_int + 2 * B, d_int + 3 * B, /*drain=*/false);
}

}  // namespace

extern "C" {

int qilqr_abi_version(void) { return QILQR_ABI_VERSION; }

const char *qilqr_last_error(void) { return g_last_error.c_str(); }

// the caller's structure (dev_bytes of it: the fields of the header it was compiled with) over the defaults
static bool read_device_config(const qilqr_device_config *dev, size_t dev_bytes, qilqr_device_config *dc) {
  *dc = qilqr_device_config{};
  dc->sync_every = 2;
  if (!dev) return true;
  if (dev_bytes < sizeof(int32_t) || dev_bytes % sizeof(int32_t) != 0) return false;
  std::memcpy(dc, dev, std::min(dev_bytes, sizeof(qilqr_device_config)));  // (a caller NEWER than the library: its extra fields are not known here)
  return true;
}

int qilqr_create(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                 int32_t n_desired, double dt_s, const qilqr_options *options,
                 const qilqr_device_config *dev, qilqr_solver **out) {
  // the symbol binaries built before ABI version 7 call: it reads the fields every such header had
  return qilqr_create_sized(model, Q, R, desired, n_desired, dt_s, options, dev, QILQR_DEVICE_CONFIG_BYTES_ABI5, out);
}

int qilqr_create_sized(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                       int32_t n_desired, double dt_s, const qilqr_options *options,
                       const qilqr_device_config *dev, size_t dev_bytes, qilqr_solver **out) {
  if (!model || !Q || !R || !options || !out || n_desired < 0 || (n_desired > 0 && !desired))
    return fail(QILQR_ERR_INVALID_ARG, "null argument");
  // QuadrotorModel ctor, quadrotor_model.cc:6-25
  ModelConsts<double> mc;
  if (!make_model_consts(model->mass_kg, model->inertia, model->arm_length_m, model->torque_to_thrust_ratio_m,
                         model->g_mpss, Q, R, dt_s, &mc))
    return fail(QILQR_ERR_BAD_INERTIA, "Inertia matrix is not positive definite!");
  if (n_desired > 0) {
    int rc = check_quaternions(desired, n_desired, "desired trajectory");
    if (rc) return rc;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(QILQR_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  qilqr_device_config dc;
  if (!read_device_config(dev, dev_bytes, &dc)) return fail(QILQR_ERR_INVALID_ARG, "dev_bytes is not the size of a qilqr_device_config");
  (void)hw_queues();  // latched here, with the process's first solver
  if (dc.device < 0 || dc.device >= ndev) return fail(QILQR_ERR_INVALID_ARG, "bad device ordinal");
  if (dc.round_launch < 0 || dc.round_launch > 2) return fail(QILQR_ERR_INVALID_ARG, "round_launch is 0 (automatic), 1 (three launches per round) or 2 (two)");
  if (!(dc.rounds_per_launch == 0 || dc.rounds_per_launch == 1 || dc.rounds_per_launch == 2 || dc.rounds_per_launch == 4))
    return fail(QILQR_ERR_INVALID_ARG, "rounds_per_launch is 0 (automatic), 1, 2 or 4");
  if (dc.sync_every < 1) dc.sync_every = 1;
#ifndef QILQR_WITH_SOLVE4
  if (dc.persistent == 1)
    return fail(QILQR_ERR_INVALID_ARG, "persistent = 1 (k_solve4, the one-launch solve) is in the diagnostics build: make -C quadrotorilqr_amd/csrc diag");
#endif
  if (dc.compaction < -1 || dc.compaction > 1) return fail(QILQR_ERR_INVALID_ARG, "compaction is -1 (never), 0 (automatic) or 1 (whenever possible)");
  if (dc.force_general == 6)
    return fail(QILQR_ERR_INVALID_ARG, "force_general = 6 (the fused k_backward4 with a block barrier per knot) was retired in round 4: 5 is the fused form");
#ifndef QILQR_WITH_BACKWARD2
  if (dc.force_general == 3)
    return fail(QILQR_ERR_INVALID_ARG, "force_general = 3 (k_backward2) is in the diagnostics build: make -C quadrotorilqr_amd/csrc diag");
#endif

  qilqr_solver *s = new qilqr_solver();
  s->device = dc.device;
  s->dev = dc;
  s->options = *options;
  s->params = SolveParams{options->step_update, options->desired_reduction_frac, options->rtol, options->atol,
                          options->max_iters, options->ls_max_iters, 0.0, 1.0, 0.0};
  s->consts = mc;
  s->f32 = (dc.precision == 1);
  convert_consts(mc, s->constsf);
  s->symmetric = true;
  for (int i = 0; i < 12; ++i)
    for (int k = 0; k < i; ++k) s->symmetric = s->symmetric && (Q[i * 12 + k] == Q[k * 12 + i]);
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < i; ++k) s->symmetric = s->symmetric && (R[i * 4 + k] == R[k * 4 + i]);
  if (dc.force_general == 1) s->symmetric = false;
  {
    bool qsym = true, ur0 = true;
    for (int i = 0; i < 12; ++i)
      for (int k = 0; k < 12; ++k) {
        qsym = qsym && (Q[i * 12 + k] == Q[k * 12 + i]);
        if (i < 6 && k >= 6) ur0 = ur0 && (Q[i * 12 + k] == 0.0);
      }
    s->layout = make_layout(qsym && dc.force_general != 1, ur0);
    bool diag = true;
    for (int i = 0; i < 12; ++i)
      for (int k = 0; k < 12; ++k)
        if (i != k) diag = diag && (Q[i * 12 + k] == 0.0);
    s->q_diag = diag && dc.dense_weights == 0;  // (qilqr_device_config.dense_weights: A/B and the bit-identity test)
  }
  s->n_desired = n_desired;

  hipError_t e = hipSetDevice(s->device);
  if (e == hipSuccess) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) == hipSuccess && cus > 0) s->num_cus = cus;
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  const size_t es = s->f32 ? sizeof(float) : sizeof(double);
  if (e == hipSuccess) e = hipMalloc(&s->d_desired, es * 18 * (n_desired > 0 ? n_desired : 1));
  if (e == hipSuccess && n_desired > 0) {
    if (s->f32) {
      std::vector<float> tmp((size_t)18 * n_desired);
      for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = (float)desired[i];
      e = hipMemcpy(s->d_desired, tmp.data(), es * tmp.size(), hipMemcpyHostToDevice);
    } else {
      e = hipMemcpy(s->d_desired, desired, es * 18 * n_desired, hipMemcpyHostToDevice);
    }
  }
  if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_counters, sizeof(int) * COUNT_WORDS, hipHostMallocDefault);
  if (e == hipSuccess)
    e = hipHostMalloc((void **)&s->h_active, sizeof(unsigned long long) * (8 * (1 + qilqr_solver::MAX_PARTS) + 1),
                      hipHostMallocMapped | hipHostMallocCoherent);  // (+ 1: the error word, BatchState::host_error)
  if (e == hipSuccess) {
    for (int k = 0; k < 8 * (1 + qilqr_solver::MAX_PARTS) + 1; ++k) s->h_active[k] = 0;
    e = hipHostGetDevicePointer((void **)&s->d_active, s->h_active, 0);
    s->st.host_active = s->d_active;
    s->st.host_error = s->d_active + 8 * (1 + qilqr_solver::MAX_PARTS);
  }
  if (e == hipSuccess) e = hipMalloc((void **)&s->d_part_counters, sizeof(int) * 2 * COUNT_WORDS * qilqr_solver::MAX_PARTS);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->main_ready, hipEventDisableTiming);
  // (the streams and events of sub-batches are created when a solve first uses them: ensure_parts)
  if (e == hipSuccess) e = hipMalloc(&s->d_consts, s->f32 ? sizeof(ModelConsts<float>) : sizeof(ModelConsts<double>));
  if (e == hipSuccess)
    e = s->f32 ? hipMemcpy(s->d_consts, &s->constsf, sizeof(ModelConsts<float>), hipMemcpyHostToDevice)
               : hipMemcpy(s->d_consts, &s->consts, sizeof(ModelConsts<double>), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&s->d_ctab, es * CTAB_SIZE);
  if (e == hipSuccess) {
    double tab[CTAB_SIZE];
    float tabf[CTAB_SIZE];
    build_ctab(s->consts.Bu, s->consts.Q, tab);
    for (int i = 0; i < CTAB_SIZE; ++i) tabf[i] = (float)tab[i];
    e = hipMemcpy(s->d_ctab, s->f32 ? (const void *)tabf : (const void *)tab, es * CTAB_SIZE, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    const int rc = fail(QILQR_ERR_HIP, std::string("qilqr_create: ") + hipGetErrorString(e));
    qilqr_destroy(s);
    return rc;
  }
  *out = s;
  return QILQR_OK;
}

void qilqr_destroy(qilqr_solver *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  free_workspace(s);
  for (auto &e : s->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  if (s->early_stream) (void)hipStreamDestroy(s->early_stream);
  if (s->early_evt) (void)hipEventDestroy(s->early_evt);
  if (s->early_done) (void)hipEventDestroy(s->early_done);
  if (s->d_early) (void)hipFree(s->d_early);
  if (s->d_late) (void)hipFree(s->d_late);
  if (s->h_late) (void)hipHostFree(s->h_late);
  if (s->dbg_trajs) (void)hipFree(s->dbg_trajs);
  if (s->dbg_cost) (void)hipFree(s->dbg_cost);
  if (s->dbg_seen) (void)hipFree(s->dbg_seen);
  if (s->stage_traj) (void)hipFree(s->stage_traj);
  if (s->stage_des) (void)hipFree(s->stage_des);
  if (s->stage_cost) (void)hipFree(s->stage_cost);
  if (s->stage_int) (void)hipFree(s->stage_int);
  if (s->d_desired) (void)hipFree(s->d_desired);
  if (s->d_ctab) (void)hipFree(s->d_ctab);
  if (s->d_consts) (void)hipFree(s->d_consts);
  if (s->h_counters) (void)hipHostFree(s->h_counters);
  if (s->h_active) (void)hipHostFree(s->h_active);
  if (s->d_part_counters) (void)hipFree(s->d_part_counters);
  if (s->main_ready) (void)hipEventDestroy(s->main_ready);
  for (int k = 0; k < qilqr_solver::MAX_PARTS; ++k) {
    if (s->part_done[k]) (void)hipEventDestroy(s->part_done[k]);
    if (s->part_stream[k]) (void)hipStreamDestroy(s->part_stream[k]);
  }
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int qilqr_device(const qilqr_solver *s) { return s ? s->device : -1; }
void *qilqr_stream(const qilqr_solver *s) { return s ? (void *)s->stream : nullptr; }

int qilqr_cost_history(qilqr_solver *s, int32_t B, double *hist, int32_t cap, int32_t *out_cap) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null solver");
  if (out_cap) *out_cap = s->hist_cap;
  if (!hist) return QILQR_OK;
  if (!s->options.populate_debug || !s->st.cost_hist || s->hist_cap <= 0)
    return fail(QILQR_ERR_INVALID_ARG, "cost history needs options.populate_debug");
  if (B <= 0 || B > s->cap_B || cap < s->hist_cap) return fail(QILQR_ERR_INVALID_ARG, "bad B or cap");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::vector<double> tmp((size_t)B * s->hist_cap);
  std::vector<int> iters(B);
  HIP_TRY(hipMemcpy(tmp.data(), s->st.cost_hist, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(iters.data(), s->st.iters, sizeof(int) * B, hipMemcpyDeviceToHost));
  for (long b = 0; b < B; ++b)
    for (int k = 0; k < cap; ++k)
      hist[b * cap + k] = (k < iters[b] && k < s->hist_cap) ? tmp[b * s->hist_cap + k] : std::nan("");
  return QILQR_OK;
}

int qilqr_profile_reset(qilqr_solver *s) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null solver");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->events_used = 0;
  for (int k = 0; k < K_KINDS; ++k) {
    s->prof_ms[k] = 0;
    s->prof_n[k] = 0;
    s->prof_seen[k] = 0;
  }
  return QILQR_OK;
}

int qilqr_profile_mode(qilqr_solver *s, int32_t mode) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null solver");
  if (mode < 0 || (mode & 0xff) > 4 || (mode >> 16)) return fail(QILQR_ERR_INVALID_ARG, "profile mode must be 0..4 (+ stride << 8)");
  int rc = qilqr_profile_reset(s);
  if (rc) return rc;
  s->dev.profile = mode;
  return QILQR_OK;
}

int qilqr_set_regularisation(qilqr_solver *s, double mu_init, double mu_factor, double mu_max) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null solver");
  if (!(mu_init >= 0.0) || !std::isfinite(mu_init)) return fail(QILQR_ERR_INVALID_ARG, "mu_init must be finite and >= 0");
  if (mu_init > 0.0) {
    if (!(mu_factor > 1.0) || !std::isfinite(mu_factor)) return fail(QILQR_ERR_INVALID_ARG, "mu_factor must be finite and > 1");
    if (!(mu_max >= mu_init) || !std::isfinite(mu_max)) return fail(QILQR_ERR_INVALID_ARG, "mu_max must be finite and >= mu_init");
    // the restarts of one iteration must fit the round bound of run_solve
    if (std::log(mu_max / mu_init) / std::log(mu_factor) > 1000.0)
      return fail(QILQR_ERR_INVALID_ARG, "more than 1000 restarts between mu_init and mu_max");
  } else {
    mu_factor = 1.0;
    mu_max = 0.0;
  }
  s->params.mu_init = mu_init;
  s->params.mu_factor = mu_factor;
  s->params.mu_max = mu_max;
  return QILQR_OK;
}

int qilqr_set_integrator(qilqr_solver *s, int32_t integrator) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null solver");
  if (integrator != 0 && integrator != 1) return fail(QILQR_ERR_INVALID_ARG, "integrator must be 0 (explicit Euler) or 1 (Runge-Kutta)");
  if (integrator == 1 && s->f32) return fail(QILQR_ERR_INVALID_ARG, "the Runge-Kutta extension needs precision 0 (fp64)");
  if (integrator == s->integrator) return QILQR_OK;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  // the knot records change shape (dense M instead of the Euler step's six blocks): the workspace is rebuilt on the next call
  free_workspace(s);
  s->integrator = integrator;
  s->layout = make_layout(s->layout.sym != 0, s->layout.ur_zero != 0, integrator == 1);
  return QILQR_OK;
}

int qilqr_profile_get(qilqr_solver *s, qilqr_profile *out) {
  if (!s || !out) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  drain_events(s);
  out->backward_ms = s->prof_ms[K_BACKWARD];
  out->backward_launches = s->prof_n[K_BACKWARD];
  out->rollout_ms = s->prof_ms[K_ROLLOUT];
  out->rollout_launches = s->prof_n[K_ROLLOUT];
  out->linearize_ms = s->prof_ms[K_LINEARIZE];
  out->linearize_launches = s->prof_n[K_LINEARIZE];
  out->other_ms = s->prof_ms[K_OTHER];
  out->other_launches = s->prof_n[K_OTHER];
  out->backward_seen = (int32_t)s->prof_seen[K_BACKWARD];
  out->rollout_seen = (int32_t)s->prof_seen[K_ROLLOUT];
  out->linearize_seen = (int32_t)s->prof_seen[K_LINEARIZE];
  out->other_seen = (int32_t)s->prof_seen[K_OTHER];
  out->solve_ms = s->prof_ms[K_SOLVE];
  out->solve_launches = s->prof_n[K_SOLVE];
  out->solve_seen = (int32_t)s->prof_seen[K_SOLVE];
  return QILQR_OK;
}

int qilqr_solve_batch_device(qilqr_solver *s, const double *d_init, const double *d_desired_batch, int32_t B,
                             int32_t n, double *d_out_traj, double *d_out_cost, int32_t *d_out_status,
                             int32_t *d_out_iters, int32_t *d_out_n_bwd, int32_t *d_out_n_fwd) {
  return solve_batch_device_impl(s, d_init, d_desired_batch, B, n, d_out_traj, d_out_cost, d_out_status, d_out_iters,
                                 d_out_n_bwd, d_out_n_fwd, /*drain=*/true);
}

// Order the solver's stream behind work of another stream: the solver's stream waits (on the device, no host
// stall) for `hip_event`, a hipEvent_t the caller recorded on the stream that produces the input buffers.
int qilqr_stream_wait_event(qilqr_solver *s, void *hip_event) {
  if (!s || !hip_event) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamWaitEvent(s->stream, (hipEvent_t)hip_event, 0));
  return QILQR_OK;
}

// host-buffer wrapper: solve_batch_staged, then the copies back -- behind the gather on the solver's stream, or, for a batch
// whose outputs are pinned, in two parts with the first under the tail rounds of the solve (EarlyOut)
namespace {
bool pinned_or_null(const void *p) {
  if (!p) return true;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an ordinary malloc'ed pointer is reported as an error by some runtimes: not pinned, and not sticky)
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
}  // namespace
int qilqr_solve_batch(qilqr_solver *s, const double *init, const double *desired_batch, int32_t B, int32_t n,
                      double *out_traj, double *out_cost, int32_t *out_status, int32_t *out_iters,
                      int32_t *out_n_bwd, int32_t *out_n_fwd) {
  if (!s) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  // the two-part copy-back pays when the trajectories are megabytes and the rounds run free on one stream; it needs pinned
  // outputs (a copy to pageable memory would hold this thread, which has rounds to enqueue)
  EarlyOut eo{out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd};
  LateLayout late{};
  eo.layout = &late;
  const bool early = out_traj && B >= 256 && (size_t)B * n * 144 >= ((size_t)2 << 20) && s->dev.sync_every > 1 && auto_parts(s, B) == 1 &&
                     !use_persistent(s, B) && 0.0 < s->params.max_iters && pinned_or_null(out_traj) && pinned_or_null(out_cost) &&
                     pinned_or_null(out_status) && pinned_or_null(out_iters) && pinned_or_null(out_n_bwd) && pinned_or_null(out_n_fwd);
  if (early) {
    eo.threshold = (unsigned)(B / 8);
    HIP_TRY(hipSetDevice(s->device));
    // (the staged form of the late part stays as the fallback for arrays that do not map; the diagnostics build can force it: A/B, its test)
    auto mapped = [](void *h, auto **v) -> bool {
      if (!h) return true;
      void *d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return false;
      }
      *v = (std::remove_reference_t<decltype(*v)>)d;
      return true;
    };
    eo.direct = !g_force_staged_late && mapped(out_traj, &eo.v_traj) && mapped(out_cost, &eo.v_cost) && mapped(out_status, &eo.v_status) &&
                mapped(out_iters, &eo.v_iters) && mapped(out_n_bwd, &eo.v_bwd) && mapped(out_n_fwd, &eo.v_fwd);
    int rc0 = ensure_early_buffers(s, B, n, eo.threshold);
    if (rc0) return rc0;
    s->early_out = &eo;
  }
  int rc = solve_batch_staged(s, init, desired_batch, B, n);
  s->early_out = nullptr;
  if (rc != QILQR_OK) {
    if (eo.fired) {  // nothing of a failed call keeps writing the caller's arrays
      (void)hipStreamSynchronize(s->early_stream);
      (void)hipStreamSynchronize(s->stream);
    }
    return rc;
  }
  const size_t tb = sizeof(double) * 18 * (size_t)B * n;
  const double *d_cost = s->stage_cost;
  const int *d_int = s->stage_int;
  hipError_t e = hipSuccess;
  if (eo.fired) {
    e = hipStreamSynchronize(s->early_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && !eo.direct) {
      const int count = *(const int *)(s->h_late + late.count);
      if (count < 0 || count > eo.late_cap) return fail(QILQR_ERR_HIP, "copy back: more late trajectories than were running");
      const int *idx = (const int *)(s->h_late + late.idx);
      const double *lt = (const double *)(s->h_late + late.traj), *lc = (const double *)(s->h_late + late.cost);
      const int *li = (const int *)(s->h_late + late.ints);
      const size_t row = (size_t)n * 18;
      for (int k = 0; k < count; ++k) {
        const int b = idx[k];
        std::memcpy(out_traj + (size_t)b * row, lt + (size_t)k * row, sizeof(double) * row);
        if (out_cost) out_cost[b] = lc[k];
        if (out_status) out_status[b] = li[k];
        if (out_iters) out_iters[b] = li[eo.late_cap + k];
        if (out_n_bwd) out_n_bwd[b] = li[2 * eo.late_cap + k];
        if (out_n_fwd) out_n_fwd[b] = li[3 * eo.late_cap + k];
      }
    }
  } else {
    if (out_traj) e = hipMemcpyAsync(out_traj, s->stage_traj, tb, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && out_cost) e = hipMemcpyAsync(out_cost, d_cost, sizeof(double) * B, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && out_status) e = hipMemcpyAsync(out_status, d_int, sizeof(int) * B, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && out_iters) e = hipMemcpyAsync(out_iters, d_int + B, sizeof(int) * B, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && out_n_bwd) e = hipMemcpyAsync(out_n_bwd, d_int + 2 * B, sizeof(int) * B, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && out_n_fwd) e = hipMemcpyAsync(out_n_fwd, d_int + 3 * B, sizeof(int) * B, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (s->dev.profile) drain_events(s);
  if (e != hipSuccess) return fail(QILQR_ERR_HIP, std::string("copy back: ") + hipGetErrorString(e));
  return device_error(s);
}

// pinned host memory for callers of the host-buffer entry points (direct DMA instead of HIP's pageable staging)
void *qilqr_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    g_last_error = "hipHostMalloc failed";
    return nullptr;
  }
  return p;
}
void qilqr_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int qilqr_solve(qilqr_solver *s, const double *init, int32_t n, double *out_traj, double *out_cost,
                int32_t *out_status, int32_t *out_iters, double *debug_cost, double *debug_trajs,
                int32_t debug_cap, int32_t *n_debug) {
  if (!s || !init || !out_traj) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  if (n <= 0) return fail(QILQR_ERR_INVALID_ARG, "empty trajectory");
  int rc;
  if ((rc = check_quaternions(init, n, "initial trajectory"))) return rc;
  if ((rc = begin_batch(s, 1, n, nullptr))) return rc;
  if ((rc = upload_tiled(s, init, s->st.traj[0], 1, n, 18))) return rc;
  const bool want_debug = s->options.populate_debug && debug_cap > 0 && (debug_cost || debug_trajs);
  if (want_debug) {
    // the ring lives in device memory (k_debug_capture appends to it behind every round's settle step); one download at the end
    const size_t want_t = debug_trajs ? (size_t)debug_cap * n * 18 : 0, want_c = (size_t)debug_cap;
    if (want_t > s->dbg_traj_cap) {
      if (s->dbg_trajs) (void)hipFree(s->dbg_trajs);
      s->dbg_trajs = nullptr;
      s->dbg_traj_cap = 0;
      HIP_TRY(hipMalloc((void **)&s->dbg_trajs, sizeof(double) * want_t));
      s->dbg_traj_cap = want_t;
    }
    if (want_c > s->dbg_cost_cap) {
      if (s->dbg_cost) (void)hipFree(s->dbg_cost);
      s->dbg_cost = nullptr;
      s->dbg_cost_cap = 0;
      HIP_TRY(hipMalloc((void **)&s->dbg_cost, sizeof(double) * want_c));
      s->dbg_cost_cap = want_c;
    }
    if (!s->dbg_seen) HIP_TRY(hipMalloc((void **)&s->dbg_seen, sizeof(int)));
    HIP_TRY(hipMemsetAsync(s->dbg_seen, 0, sizeof(int), s->stream));
    // the ring as k_round sees it: an idle wavefront of the launch captures behind every round's backward pass (debug_capture_wave), so the
    // launches keep their four rounds; rounds of separate launches are followed by k_debug_capture as before (`capture` below)
    s->st.dbg_trajs = debug_trajs ? s->dbg_trajs : nullptr;
    s->st.dbg_cost = s->dbg_cost;
    s->st.dbg_seen = s->dbg_seen;
    s->st.dbg_cap = (int)debug_cap;
  }
  struct DebugRingScope {  // (no other entry point sees the ring)
    qilqr_solver *s;
    ~DebugRingScope() { s->st.dbg_trajs = s->st.dbg_cost = nullptr; s->st.dbg_seen = nullptr; s->st.dbg_cap = 0; }
  } ring_scope{s};
  auto capture = [&]() -> int {
    // ilqr.hh:78-80: one entry per completed forward pass (accepted iteration)
    if (!want_debug || s->round_captured) return QILQR_OK;  // (a k_round launch has captured its own rounds)
    if (s->f32)
      launch(s, K_OTHER, k_debug_capture<float>, dim3(1), dim3(256), s->st, (int)n, debug_trajs ? s->dbg_trajs : nullptr, s->dbg_cost, s->dbg_seen, (int)debug_cap);
    else
      launch(s, K_OTHER, k_debug_capture<double>, dim3(1), dim3(256), s->st, (int)n, debug_trajs ? s->dbg_trajs : nullptr, s->dbg_cost, s->dbg_seen, (int)debug_cap);
    return QILQR_OK;
  };
  // (without debug entries nobody looks at the solve round by round: the launches may hold several rounds)
  if ((rc = run_solve(s, 1, n, s->dev.sync_every, capture, true, nullptr, /*double_ok=*/true))) return rc;
  int status = 0, iters = 0, seen = 0;
  double cost = 0;
  HIP_TRY(hipMemcpy(&status, s->st.status, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&iters, s->st.iters, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&cost, s->st.cost, sizeof(double), hipMemcpyDeviceToHost));
  if (want_debug) {
    HIP_TRY(hipMemcpy(&seen, s->dbg_seen, sizeof(int), hipMemcpyDeviceToHost));
    const int have = seen < debug_cap ? seen : debug_cap;
    if (have > 0 && debug_cost) HIP_TRY(hipMemcpy(debug_cost, s->dbg_cost, sizeof(double) * have, hipMemcpyDeviceToHost));
    if (have > 0 && debug_trajs) HIP_TRY(hipMemcpy(debug_trajs, s->dbg_trajs, sizeof(double) * (size_t)have * n * 18, hipMemcpyDeviceToHost));
  }
  if (s->dev.profile) drain_events(s);
  if (n_debug) *n_debug = want_debug ? (seen < debug_cap ? seen : debug_cap) : 0;
  if (status == QILQR_STATUS_LINE_SEARCH_FAILED)
    return fail(QILQR_ERR_LINE_SEARCH, "Reached maximum number of line search iterations, " +
                                           std::to_string(s->options.ls_max_iters) + "\n");
  if ((rc = download_tiled(s, out_traj, s->st.traj[0], s->st.traj[1], s->st.cur, 0, 1, n, 18))) return rc;
  if (out_cost) *out_cost = cost;
  if (out_status) *out_status = status;
  if (out_iters) *out_iters = iters;
  return QILQR_OK;
}

int qilqr_cost_trajectory(qilqr_solver *s, const double *traj, int32_t B, int32_t n, double *cost) {
  if (!s || !traj || !cost) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  int rc = begin_batch(s, B, n, nullptr);
  if (rc) return rc;
  if ((rc = upload_tiled(s, traj, s->st.traj[0], B, n, 18))) return rc;
  if ((rc = launch_linearize(s, B, n, 0, 0))) return rc;
  launch(s, K_OTHER, k_init, dim3(cdiv(B, 64)), dim3(64), s->params, s->st, (int)B, (int)n);
  HIP_TRY(hipMemcpyAsync(cost, s->st.cost, sizeof(double) * B, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipGetLastError());
  return QILQR_OK;
}

int qilqr_backwards_pass(qilqr_solver *s, const double *traj, int32_t B, int32_t n, double *gains, double *terms) {
  if (!s || !traj || !gains || !terms) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  int rc = begin_batch(s, B, n, nullptr);
  if (rc) return rc;
  if ((rc = upload_tiled(s, traj, s->st.traj[0], B, n, 18))) return rc;
  if ((rc = launch_linearize(s, B, n, 0, 0))) return rc;
  launch(s, K_OTHER, k_init, dim3(cdiv(B, 64)), dim3(64), s->params, s->st, (int)B, (int)n);
  if ((rc = launch_backward(s, B, n, 1))) return rc;
  if ((rc = download_tiled(s, gains, s->st.gains, s->st.gains, nullptr, 0, B, n, 52))) return rc;
  HIP_TRY(hipMemcpyAsync(terms, s->st.terms, sizeof(double) * 2 * B, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipGetLastError());
  return device_error(s);
}

int qilqr_forward_sim(qilqr_solver *s, const double *traj, const double *gains, const double *alpha, int32_t B,
                      int32_t n, double *out_traj) {
  if (!s || !traj || !gains || !alpha || !out_traj) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  int rc = begin_batch(s, B, n, nullptr);
  if (rc) return rc;
  if ((rc = upload_tiled(s, traj, s->st.traj[0], B, n, 18))) return rc;
  if ((rc = upload_tiled(s, gains, s->st.gains, B, n, 52))) return rc;
  HIP_TRY(hipMemcpyAsync(s->st.alpha, alpha, sizeof(double) * B, hipMemcpyHostToDevice, s->stream));
  if ((rc = launch_rollout(s, B, n, 0))) return rc;
  if ((rc = download_tiled(s, out_traj, s->st.traj[1], s->st.traj[1], nullptr, 0, B, n, 18))) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipGetLastError());
  return device_error(s);
}

int qilqr_line_search(qilqr_solver *s, const double *traj, const double *cost, const double *gains,
                      const double *terms, int32_t B, int32_t n, double *out_traj, double *out_cost,
                      double *out_step, int32_t *out_status) {
  if (!s || !traj || !cost || !gains || !terms) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  int rc = begin_batch(s, B, n, nullptr);
  if (rc) return rc;
  struct Scratch {  // freed on every return path
    double *cost = nullptr, *terms = nullptr;
    ~Scratch() {
      if (cost) (void)hipFree(cost);
      if (terms) (void)hipFree(terms);
    }
  } scratch;
  HIP_TRY(hipMalloc((void **)&scratch.cost, sizeof(double) * B));
  HIP_TRY(hipMalloc((void **)&scratch.terms, sizeof(double) * 2 * B));
  double *const d_cost = scratch.cost, *const d_terms = scratch.terms;
  if ((rc = upload_tiled(s, traj, s->st.traj[0], B, n, 18))) return rc;
  if ((rc = upload_tiled(s, gains, s->st.gains, B, n, 52))) return rc;
  HIP_TRY(hipMemcpyAsync(d_cost, cost, sizeof(double) * B, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(d_terms, terms, sizeof(double) * 2 * B, hipMemcpyHostToDevice, s->stream));
  launch(s, K_OTHER, k_seed_search, dim3(cdiv(B, 64)), dim3(64), s->st, (int)B, d_cost, d_terms);
  if (s->params.ls_max_iters <= 0) {
    // ilqr.hh:178: the loop body never runs, the reference throws at once
    std::vector<int> st3(B, QILQR_STATUS_LINE_SEARCH_FAILED);
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (out_status) std::memcpy(out_status, st3.data(), sizeof(int) * B);
    return QILQR_OK;
  }
  for (int t = 0; t < s->params.ls_max_iters; ++t) {
    HIP_TRY(hipMemsetAsync(s->st.counters, 0, sizeof(int) * COUNT_WORDS, s->stream));
    if ((rc = launch_rollout(s, B, n, F_SEARCH))) return rc;
    if ((rc = launch_linearize(s, B, n, 1, F_SEARCH))) return rc;
    if ((rc = launch_accept(s, B, n, 1))) return rc;
    int n_active = 0;
    if ((rc = read_active(s, &n_active))) return rc;
    if (n_active == 0) break;
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  // results: accepted candidates are traj[cur] (cur flipped); failures keep the input
  std::vector<int> status(B);
  HIP_TRY(hipMemcpy(status.data(), s->st.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (out_status) std::memcpy(out_status, status.data(), sizeof(int) * B);
  if (out_cost) HIP_TRY(hipMemcpy(out_cost, s->st.cost, sizeof(double) * B, hipMemcpyDeviceToHost));
  if (out_step) HIP_TRY(hipMemcpy(out_step, s->st.alpha, sizeof(double) * B, hipMemcpyDeviceToHost));
  if (out_traj && (rc = download_tiled(s, out_traj, s->st.traj[0], s->st.traj[1], s->st.cur, 0, B, n, 18))) return rc;
  HIP_TRY(hipGetLastError());
  return device_error(s);
}

// ---- one batch over several devices in one process (include/quadrotor_ilqr.h)
// RCCL is bound at run time (dlopen of librccl.so.1 when a sharded handle first needs a communicator), not as a link-time
// dependency: the library is 570 MB of code objects that a single-device caller never uses, and a host process that has
// PyTorch loaded already holds one under the same soname, which is then the one that is used.
extern "C++" {
namespace {
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error, path;
  bool ok = false;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // Which librccl: the one that ships BESIDE the HIP runtime this process runs on (found from the address of a HIP entry
    // point).  A process can hold two ROCm installations -- PyTorch bundles its own libamdhip64 and librccl, this library is
    // linked against /opt/rocm's, and whichever libamdhip64 was loaded first serves both -- and an RCCL built for the other
    // runtime fails in ncclCommInitAll ("unhandled cuda error": measured, torch's librccl 7.0 on the 7.2 runtime).  Loaded
    // RTLD_LOCAL: a second copy beside one the host process already uses keeps its own state and exports nothing.
    std::vector<std::string> names;
    Dl_info where;
    if (dladdr((const void *)&hipGetDeviceCount, &where) && where.dli_fname) {
      const std::string path(where.dli_fname);
      const size_t slash = path.rfind('/');
      if (slash != std::string::npos) {
        names.push_back(path.substr(0, slash + 1) + "librccl.so.1");
        names.push_back(path.substr(0, slash + 1) + "librccl.so");
      }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string &name : names) {
      r.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.lib) {
        r.path = name;
        break;
      }
    }
    if (!r.lib) {
      const char *err = dlerror();
      r.error = std::string("librccl.so.1 cannot be loaded: ") + (err ? err : "?");
      return;
    }
#define QILQR_RCCL_SYM(field, sym)                                    \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym)); \
  if (!r.field) {                                                     \
    r.error = "librccl has no " #sym;                                 \
    return;                                                           \
  }
    QILQR_RCCL_SYM(GetVersion, ncclGetVersion)
    QILQR_RCCL_SYM(CommInitAll, ncclCommInitAll)
    QILQR_RCCL_SYM(CommDestroy, ncclCommDestroy)
    QILQR_RCCL_SYM(GroupStart, ncclGroupStart)
    QILQR_RCCL_SYM(GroupEnd, ncclGroupEnd)
    QILQR_RCCL_SYM(Send, ncclSend)
    QILQR_RCCL_SYM(Recv, ncclRecv)
    QILQR_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef QILQR_RCCL_SYM
    r.ok = true;
  });
  return r;
}
// the caller's current HIP device is put back on every return path (a torch host process has one)
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};
}  // namespace
}  // extern "C++"

struct qilqr_sharded {
  std::vector<qilqr_solver *> solvers;
  std::vector<int> device;     // per shard: its HIP device
  std::vector<int> uniq;       // the distinct devices, in order of first appearance: rank k of the communicator is uniq[k]
  std::vector<int> rank_of;    // per shard: index of its device in uniq
  int transport = QILQR_TRANSPORT_AUTO;  // as requested
  int resolved = 0;                      // QILQR_TRANSPORT_RCCL or QILQR_TRANSPORT_PEER_COPY once the gather path exists
  std::vector<ncclComm_t> comms;         // per distinct device (RCCL transport)
  std::vector<hipStream_t> gstream;      // per distinct device: the stream the gather of that device's shards runs on
  std::vector<hipEvent_t> done;          // per shard: its solve (and gather kernel) have finished
  std::mutex gather_mutex;               // the shards' host threads enqueue their transfers one at a time (shared communicators / streams)
  std::string info;
};

extern "C++" {
namespace {
void sharded_drop_gather_path(qilqr_sharded *h) {
  for (size_t k = 0; k < h->comms.size(); ++k)
    if (h->comms[k]) (void)rccl().CommDestroy(h->comms[k]);
  h->comms.clear();
  h->resolved = 0;
}
// communicators (RCCL) or nothing (peer copies), streams and events: built when the first gathered solve needs them
int sharded_ensure_gather_path(qilqr_sharded *h) {
  if (h->resolved) return QILQR_OK;
  const int nuniq = (int)h->uniq.size();
  if (h->gstream.empty()) {
    h->gstream.assign(nuniq, nullptr);
    for (int k = 0; k < nuniq; ++k) {
      HIP_TRY(hipSetDevice(h->uniq[k]));
      HIP_TRY(hipStreamCreateWithFlags(&h->gstream[k], hipStreamNonBlocking));
    }
    h->done.assign(h->solvers.size(), nullptr);
    for (size_t r = 0; r < h->solvers.size(); ++r) {
      HIP_TRY(hipSetDevice(h->device[r]));
      HIP_TRY(hipEventCreateWithFlags(&h->done[r], hipEventDisableTiming));
    }
  }
  // automatic: RCCL when the shards sit on more than one device (point-to-point over xGMI into the root's rows), plain device
  // copies when they all share one (nothing to communicate)
  int want = h->transport;
  if (want == QILQR_TRANSPORT_AUTO) want = nuniq > 1 ? QILQR_TRANSPORT_RCCL : QILQR_TRANSPORT_PEER_COPY;
  if (want == QILQR_TRANSPORT_RCCL) {
    Rccl &R = rccl();
    if (!R.ok) {
      if (h->transport == QILQR_TRANSPORT_RCCL) return fail(QILQR_ERR_HIP, "RCCL transport requested: " + R.error);
      want = QILQR_TRANSPORT_PEER_COPY;
      h->info = "peer copies (hipMemcpyPeerAsync): " + R.error;
    } else {
      h->comms.assign(nuniq, nullptr);
      const ncclResult_t st = R.CommInitAll(h->comms.data(), nuniq, h->uniq.data());
      if (st != ncclSuccess) {
        h->comms.clear();
        const std::string why = std::string("ncclCommInitAll: ") + R.GetErrorString(st);
        if (h->transport == QILQR_TRANSPORT_RCCL) return fail(QILQR_ERR_HIP, "RCCL transport requested: " + why);
        want = QILQR_TRANSPORT_PEER_COPY;
        h->info = "peer copies (hipMemcpyPeerAsync): " + why;
      } else {
        int ver = 0;
        (void)R.GetVersion(&ver);
        h->info = "rccl: ncclSend / ncclRecv, " + std::to_string(nuniq) + " rank" + (nuniq > 1 ? "s" : "") + " in one process (version code " +
                  std::to_string(ver) + ", " + R.path + ")";
      }
    }
  }
  if (want == QILQR_TRANSPORT_PEER_COPY && h->info.rfind("peer copies", 0) != 0) h->info = "peer copies (hipMemcpyPeerAsync)";
  h->resolved = want;
  return QILQR_OK;
}

// One shard's solve on the calling thread.  host_out: copy back into the caller's host arrays (qilqr_solve_batch); otherwise
// leave the results in the solver's staging buffers and record done[r] behind them (qilqr_solve_batch_sharded_device).
struct ShardCall {
  const double *init, *desired;
  int32_t B, n;
  double *out_traj, *out_cost;
  int32_t *out_status, *out_iters, *out_n_bwd, *out_n_fwd;
  bool host_out;
  int32_t root = -1;  // !host_out: the out_* arrays are device arrays on devices[root]; every shard sends its rows there itself
};
// One transfer of the gather: `count` elements of array `array` (0 traj, 1 cost, 2 status, 3 iters, 4 n_bwd, 5 n_fwd) from
// element src_off of shard `shard`'s staging buffer (rank src_rank of the communicator = the shard's device) to element dst_off
// of the root's array (rank dst_rank).  The whole schedule is a function of (B, n, shards, root, which arrays, ranks) alone --
// qilqr_gather_schedule below exposes it so that a test can check it for eight distinct devices without any.
struct GatherPiece {
  int32_t shard, array, src_rank, dst_rank;
  int64_t src_off, dst_off, count;
};
inline void gather_schedule_of_shard(int32_t B, int32_t n, int32_t k, int32_t r, int32_t root, const int *rank_of, unsigned arrays,
                                     std::vector<GatherPiece> *out) {
  int32_t b0 = 0, cnt = 0;
  (void)qilqr_shard_range(B, k, r, &b0, &cnt);
  if (cnt == 0) return;
  const int64_t row = (int64_t)n * 18;
  if (arrays & 1u) out->push_back({r, 0, rank_of[r], rank_of[root], 0, (int64_t)b0 * row, (int64_t)cnt * row});
  if (arrays & 2u) out->push_back({r, 1, rank_of[r], rank_of[root], 0, (int64_t)b0, (int64_t)cnt});
  for (int q = 0; q < 4; ++q)  // the four int32 arrays sit one behind the other in the shard's staging block: [4][cnt]
    if (arrays & (4u << q)) out->push_back({r, 2 + q, rank_of[r], rank_of[root], (int64_t)q * cnt, (int64_t)b0, (int64_t)cnt});
}
// enqueue shard r's transfers behind its solve (called from the shard's own host thread as soon as the solve is enqueued and
// its completion event recorded: the rows travel while slower shards still solve).  RCCL: ONE group per shard -- a send on the
// shard's communicator and a receive on the root's for each piece; communicators and gather streams are shared between
// shards (every shard receives on the root's), so the enqueue is serialised by the handle's mutex.
int enqueue_shard_gather(qilqr_sharded *h, const ShardCall &c, int32_t r);
void run_shard(qilqr_sharded *h, const ShardCall &c, int32_t r, int *rc_out, std::string *msg_out) {
  const int32_t k = (int32_t)h->solvers.size();
  int32_t b0 = 0, cnt = 0;
  (void)qilqr_shard_range(c.B, k, r, &b0, &cnt);
  if (cnt == 0) return;  // fewer problems than shards
  const size_t to = (size_t)b0 * c.n * 18;
  int rc;
  if (c.host_out) {
    rc = qilqr_solve_batch(h->solvers[r], c.init + to, c.desired ? c.desired + to : nullptr, cnt, c.n,
                           c.out_traj ? c.out_traj + to : nullptr, c.out_cost ? c.out_cost + b0 : nullptr,
                           c.out_status ? c.out_status + b0 : nullptr, c.out_iters ? c.out_iters + b0 : nullptr,
                           c.out_n_bwd ? c.out_n_bwd + b0 : nullptr, c.out_n_fwd ? c.out_n_fwd + b0 : nullptr);
  } else {
    rc = solve_batch_staged(h->solvers[r], c.init + to, c.desired ? c.desired + to : nullptr, cnt, c.n);
    if (rc == QILQR_OK && hipEventRecord(h->done[r], h->solvers[r]->stream) != hipSuccess) rc = fail(QILQR_ERR_HIP, "hipEventRecord");
    if (rc == QILQR_OK && c.root >= 0) rc = enqueue_shard_gather(h, c, r);
  }
  *rc_out = rc;
  if (rc != QILQR_OK) *msg_out = g_last_error;  // (thread-local: carried back to the caller)
}
// every shard on a host thread of its own (HIP's current device is per thread), shard 0 on the caller's
int run_all_shards(qilqr_sharded *h, const ShardCall &c) {
  const int32_t k = (int32_t)h->solvers.size();
  std::vector<int> rcs(k, QILQR_OK);
  std::vector<std::string> msgs(k);
  std::vector<std::thread> threads;
  bool spawn_failed = false;
  try {
    threads.reserve(k);
    for (int32_t r = 1; r < k; ++r) threads.emplace_back(run_shard, h, std::cref(c), r, &rcs[r], &msgs[r]);
  } catch (...) {
    spawn_failed = true;  // (std::system_error: no thread could be started; the shards that have one still run)
  }
  const int32_t started = 1 + (int32_t)threads.size();
  run_shard(h, c, 0, &rcs[0], &msgs[0]);
  for (std::thread &t : threads) t.join();
  if (spawn_failed)  // the shards without a thread run here, one after the other
    for (int32_t r = started; r < k; ++r) run_shard(h, c, r, &rcs[r], &msgs[r]);
  for (int32_t r = 0; r < k; ++r)
    if (rcs[r] != QILQR_OK)
      return fail(rcs[r], "shard " + std::to_string(r) + " (device " + std::to_string(h->device[r]) + "): " + msgs[r]);
  return QILQR_OK;
}
}  // namespace
}  // extern "C++"

int qilqr_shard_range(int32_t B, int32_t n_shards, int32_t r, int32_t *begin, int32_t *count) {
  if (B < 0 || n_shards <= 0 || r < 0 || r >= n_shards || !begin || !count) return fail(QILQR_ERR_INVALID_ARG, "bad shard arguments");
  const int32_t q = B / n_shards, rem = B % n_shards;
  *begin = r * q + (r < rem ? r : rem);
  *count = q + (r < rem ? 1 : 0);
  return QILQR_OK;
}

int qilqr_sharded_create(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                         int32_t n_desired, double dt_s, const qilqr_options *options, const qilqr_device_config *dev,
                         const int32_t *devices, int32_t n_devices, qilqr_sharded **out) {
  return qilqr_sharded_create_sized(model, Q, R, desired, n_desired, dt_s, options, dev, QILQR_DEVICE_CONFIG_BYTES_ABI5, devices, n_devices, out);
}

int qilqr_sharded_create_sized(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                               int32_t n_desired, double dt_s, const qilqr_options *options, const qilqr_device_config *dev,
                               size_t dev_bytes, const int32_t *devices, int32_t n_devices, qilqr_sharded **out) {
  if (!out || !devices || n_devices <= 0 || n_devices > 64) return fail(QILQR_ERR_INVALID_ARG, "bad device list");
  DeviceGuard guard;
  qilqr_sharded *h = nullptr;
  try {
    h = new qilqr_sharded();
    for (int32_t r = 0; r < n_devices; ++r) {
      qilqr_device_config dc;
      if (!read_device_config(dev, dev_bytes, &dc)) {
        qilqr_sharded_destroy(h);
        return fail(QILQR_ERR_INVALID_ARG, "dev_bytes is not the size of a qilqr_device_config");
      }
      dc.device = devices[r];
      qilqr_solver *s = nullptr;
      const int rc = qilqr_create_sized(model, Q, R, desired, n_desired, dt_s, options, &dc, sizeof(dc), &s);
      if (rc != QILQR_OK) {
        const std::string msg = "shard " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + g_last_error;
        qilqr_sharded_destroy(h);
        return fail(rc, msg);
      }
      h->solvers.push_back(s);
      h->device.push_back(devices[r]);
      int idx = -1;
      for (size_t u = 0; u < h->uniq.size(); ++u)
        if (h->uniq[u] == devices[r]) idx = (int)u;
      if (idx < 0) {
        idx = (int)h->uniq.size();
        h->uniq.push_back(devices[r]);
      }
      h->rank_of.push_back(idx);
    }
  } catch (...) {  // std::bad_alloc and friends do not cross the C ABI
    if (h) qilqr_sharded_destroy(h);
    return fail(QILQR_ERR_INVALID_ARG, "qilqr_sharded_create: out of host memory");
  }
  *out = h;
  return QILQR_OK;
}

int qilqr_sharded_create_mask_sized(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                                    int32_t n_desired, double dt_s, const qilqr_options *options,
                                    const qilqr_device_config *dev, size_t dev_bytes, uint64_t device_mask, qilqr_sharded **out) {
  int32_t devices[64];
  int32_t k = 0;
  for (int32_t d = 0; d < 64; ++d)
    if (device_mask & (1ull << d)) devices[k++] = d;
  if (k == 0) return fail(QILQR_ERR_INVALID_ARG, "empty device mask");
  return qilqr_sharded_create_sized(model, Q, R, desired, n_desired, dt_s, options, dev, dev_bytes, devices, k, out);
}
// (the raw symbol, for binaries built before ABI version 7: the 32 bytes of ABI version 5 -- fields behind them, `compaction` of version 6
// included, keep their defaults: the header says so)
int qilqr_sharded_create_mask(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                              int32_t n_desired, double dt_s, const qilqr_options *options,
                              const qilqr_device_config *dev, uint64_t device_mask, qilqr_sharded **out) {
  return qilqr_sharded_create_mask_sized(model, Q, R, desired, n_desired, dt_s, options, dev, QILQR_DEVICE_CONFIG_BYTES_ABI5, device_mask, out);
}

void qilqr_sharded_destroy(qilqr_sharded *h) {
  if (!h) return;
  DeviceGuard guard;
  for (qilqr_solver *s : h->solvers) qilqr_destroy(s);
  sharded_drop_gather_path(h);
  for (size_t k = 0; k < h->gstream.size(); ++k)
    if (h->gstream[k]) {
      (void)hipSetDevice(h->uniq[k]);
      (void)hipStreamDestroy(h->gstream[k]);
    }
  for (size_t r = 0; r < h->done.size(); ++r)
    if (h->done[r]) {
      (void)hipSetDevice(h->device[r]);
      (void)hipEventDestroy(h->done[r]);
    }
  delete h;
}

int32_t qilqr_sharded_count(const qilqr_sharded *h) { return h ? (int32_t)h->solvers.size() : 0; }

qilqr_solver *qilqr_sharded_solver(qilqr_sharded *h, int32_t r) {
  return (h && r >= 0 && r < (int32_t)h->solvers.size()) ? h->solvers[r] : nullptr;
}

int qilqr_solve_batch_sharded(qilqr_sharded *h, const double *init, const double *desired_batch, int32_t B, int32_t n,
                              double *out_traj, double *out_cost, int32_t *out_status, int32_t *out_iters,
                              int32_t *out_n_bwd, int32_t *out_n_fwd) {
  if (!h || h->solvers.empty() || !init) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || n <= 0) return fail(QILQR_ERR_INVALID_ARG, "B and n must be positive");
  DeviceGuard guard;
  try {
    const ShardCall c{init, desired_batch, B, n, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd, true};
    return run_all_shards(h, c);
  } catch (...) {
    return fail(QILQR_ERR_INVALID_ARG, "qilqr_solve_batch_sharded: out of host memory");
  }
}

int qilqr_sharded_set_transport(qilqr_sharded *h, int32_t transport) {
  if (!h) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  if (transport != QILQR_TRANSPORT_AUTO && transport != QILQR_TRANSPORT_RCCL && transport != QILQR_TRANSPORT_PEER_COPY)
    return fail(QILQR_ERR_INVALID_ARG, "transport must be QILQR_TRANSPORT_AUTO, _RCCL or _PEER_COPY");
  DeviceGuard guard;
  if (transport != h->transport) sharded_drop_gather_path(h);
  h->transport = transport;
  h->info.clear();
  return sharded_ensure_gather_path(h);
}

const char *qilqr_sharded_transport(qilqr_sharded *h) {
  if (!h) return "";
  DeviceGuard guard;
  if (!h->resolved && sharded_ensure_gather_path(h) != QILQR_OK) return "";
  return h->info.c_str();
}

// The C counterpart of quadrotorilqr_amd/sharding.gather_to_root: every shard's results go from the staging buffers of its
// solver straight into ITS rows of the root device's arrays -- ragged (shards differ by one problem when the count does not
// divide B), no padding, nothing concatenated afterwards.  RCCL transport: one ncclSend on the shard's device and one
// ncclRecv on the root's per array, all inside one group (a single host thread drives every communicator of the process);
// a shard on the root's own device is a send and a receive on the same communicator.  Peer-copy transport:
// hipMemcpyPeerAsync.  Either way the transfer of a shard starts when that shard's solve has finished (stream-ordered
// behind its `done` event), not when the slowest has.
extern "C++" {
namespace {
int enqueue_shard_gather(qilqr_sharded *h, const ShardCall &c, int32_t r) {
  const int32_t k = (int32_t)h->solvers.size();
  const unsigned arrays = (c.out_traj ? 1u : 0u) | (c.out_cost ? 2u : 0u) | (c.out_status ? 4u : 0u) | (c.out_iters ? 8u : 0u) |
                          (c.out_n_bwd ? 16u : 0u) | (c.out_n_fwd ? 32u : 0u);
  std::vector<GatherPiece> sched;
  gather_schedule_of_shard(c.B, c.n, k, r, c.root, h->rank_of.data(), arrays, &sched);
  if (sched.empty()) return QILQR_OK;
  qilqr_solver *s = h->solvers[r];
  const int rr = h->rank_of[c.root], root_dev = h->device[c.root];
  hipStream_t gs = h->gstream[h->rank_of[r]], rs = h->gstream[rr];
  void *dst_base[6] = {c.out_traj, c.out_cost, c.out_status, c.out_iters, c.out_n_bwd, c.out_n_fwd};
  const void *src_base[6] = {s->stage_traj, s->stage_cost, s->stage_int, s->stage_int, s->stage_int, s->stage_int};
  if (h->resolved == QILQR_TRANSPORT_RCCL) {
    // A communicator executes its operations in the order they were issued, and every shard's receives are operations of the ROOT's
    // communicator: issued when the solves are enqueued, the group of a slow shard would hold back the receives of every faster shard
    // issued behind it.  The shard's host thread therefore waits for its own solve first and issues its group then: the
    // groups reach the root's communicator in the order the shards finish.  (Peer copies run on a stream per source device and need no
    // such care.)
    HIP_TRY(hipSetDevice(h->device[r]));
    HIP_TRY(hipEventSynchronize(h->done[r]));
  }
  std::lock_guard<std::mutex> lock(h->gather_mutex);
  HIP_TRY(hipSetDevice(h->device[r]));
  HIP_TRY(hipStreamWaitEvent(gs, h->done[r], 0));
  if (h->resolved == QILQR_TRANSPORT_RCCL) {
    // the receives run on the root's gather stream: it must not start them before this shard's rows exist either
    if (rs != gs) {
      HIP_TRY(hipSetDevice(root_dev));
      HIP_TRY(hipStreamWaitEvent(rs, h->done[r], 0));
      HIP_TRY(hipSetDevice(h->device[r]));
    }
    Rccl &R = rccl();
    ncclResult_t st = R.GroupStart();
    for (const GatherPiece &pc : sched) {
      if (st != ncclSuccess) break;
      const size_t es = pc.array < 2 ? sizeof(double) : sizeof(int32_t);
      const ncclDataType_t ty = pc.array < 2 ? ncclDouble : ncclInt32;
      st = R.Send((const char *)src_base[pc.array] + es * pc.src_off, (size_t)pc.count, ty, pc.dst_rank, h->comms[pc.src_rank], gs);
      if (st == ncclSuccess) st = R.Recv((char *)dst_base[pc.array] + es * pc.dst_off, (size_t)pc.count, ty, pc.src_rank, h->comms[pc.dst_rank], rs);
    }
    const ncclResult_t st_end = R.GroupEnd();
    if (st == ncclSuccess) st = st_end;
    if (st != ncclSuccess) return fail(QILQR_ERR_HIP, std::string("RCCL gather: ") + R.GetErrorString(st));
  } else {
    for (const GatherPiece &pc : sched) {
      const size_t es = pc.array < 2 ? sizeof(double) : sizeof(int32_t);
      HIP_TRY(hipMemcpyPeerAsync((char *)dst_base[pc.array] + es * pc.dst_off, root_dev, (const char *)src_base[pc.array] + es * pc.src_off,
                                 h->device[r], es * (size_t)pc.count, gs));
    }
  }
  return QILQR_OK;
}
}  // namespace
}  // extern "C++"

// The transfers qilqr_solve_batch_sharded_device issues, without issuing them (no device is touched): for a batch of B problems of
// n knots over n_shards shards whose devices are `devices` (ordinals; equal ordinals share a communicator rank, numbered in order
// of first appearance), root shard `root`, arrays = bit mask (1 traj, 2 cost, 4 status, 8 iters, 16 n_bwd, 32 n_fwd).  out:
// 7 x int64 per transfer = {shard, array, src_rank, dst_rank, src_off, dst_off, count} in the order they are enqueued per shard
// (shard by shard here; at run time every shard's group goes out when that shard's solve has).  Returns the number of transfers
// (out may be NULL or shorter: cap entries are written).
int qilqr_gather_schedule(int32_t B, int32_t n, const int32_t *devices, int32_t n_shards, int32_t root, uint32_t arrays, int64_t *out,
                          int32_t cap) {
  if (B <= 0 || n <= 0 || n_shards <= 0 || !devices || root < 0 || root >= n_shards) return fail(QILQR_ERR_INVALID_ARG, "bad argument"), -1;
  try {
    std::vector<int> uniq, rank_of(n_shards);
    for (int32_t r = 0; r < n_shards; ++r) {
      size_t u = 0;
      while (u < uniq.size() && uniq[u] != devices[r]) ++u;
      if (u == uniq.size()) uniq.push_back(devices[r]);
      rank_of[r] = (int)u;
    }
    std::vector<GatherPiece> sched;
    for (int32_t r = 0; r < n_shards; ++r) gather_schedule_of_shard(B, n, n_shards, r, root, rank_of.data(), arrays, &sched);
    for (size_t i = 0; i < sched.size() && out && (int32_t)i < cap; ++i) {
      const GatherPiece &pc = sched[i];
      const int64_t row[7] = {pc.shard, pc.array, pc.src_rank, pc.dst_rank, pc.src_off, pc.dst_off, pc.count};
      for (int q = 0; q < 7; ++q) out[7 * i + q] = row[q];
    }
    return (int)sched.size();
  } catch (...) {
    return fail(QILQR_ERR_INVALID_ARG, "out of host memory"), -1;
  }
}

int qilqr_solve_batch_sharded_device(qilqr_sharded *h, const double *init, const double *desired_batch, int32_t B, int32_t n,
                                     int32_t root, double *d_out_traj, double *d_out_cost, int32_t *d_out_status,
                                     int32_t *d_out_iters, int32_t *d_out_n_bwd, int32_t *d_out_n_fwd, double *gather_ms) {
  if (!h || h->solvers.empty() || !init) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  if (B <= 0 || n <= 0) return fail(QILQR_ERR_INVALID_ARG, "B and n must be positive");
  const int32_t k = (int32_t)h->solvers.size();
  if (root < 0 || root >= k) return fail(QILQR_ERR_INVALID_ARG, "root must be a shard index");
  DeviceGuard guard;
  try {
    int rc = sharded_ensure_gather_path(h);
    if (rc) return rc;
    ShardCall c{init, desired_batch, B, n, d_out_traj, d_out_cost, d_out_status, d_out_iters, d_out_n_bwd, d_out_n_fwd, false};
    c.root = root;
    auto settle_everything = [&]() {  // nothing of a failed call stays in flight: solver streams AND gather streams
      for (qilqr_solver *s : h->solvers) {
        (void)hipSetDevice(s->device);
        (void)hipStreamSynchronize(s->stream);
        if (s->early_stream) (void)hipStreamSynchronize(s->early_stream);
      }
      for (size_t u = 0; u < h->uniq.size(); ++u) {
        (void)hipSetDevice(h->uniq[u]);
        (void)hipStreamSynchronize(h->gstream[u]);
      }
    };
    if ((rc = run_all_shards(h, c))) {  // (every shard has enqueued its own transfers behind its solve: enqueue_shard_gather)
      settle_everything();
      return rc;
    }
    std::vector<std::vector<int>> pieces(k);  // (only whether a shard has rows)
    for (int32_t r = 0; r < k; ++r) {
      int32_t b0 = 0, cnt = 0;
      (void)qilqr_shard_range(B, k, r, &b0, &cnt);
      if (cnt > 0) pieces[r].push_back(1);
    }
    // exposed gather time: from the moment the SLOWEST solve has finished to the moment the root holds every row (the faster
    // shards' rows have been travelling since their own solves finished)
    auto sync_or_settle = [&](hipError_t e, const char *what) -> int {
      if (e == hipSuccess) return QILQR_OK;
      settle_everything();
      return fail(QILQR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    for (int32_t r = 0; r < k; ++r)
      if (!pieces[r].empty()) {
        if ((rc = sync_or_settle(hipSetDevice(h->device[r]), "hipSetDevice"))) return rc;
        if ((rc = sync_or_settle(hipEventSynchronize(h->done[r]), "hipEventSynchronize"))) return rc;
      }
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t u = 0; u < h->uniq.size(); ++u) {
      if ((rc = sync_or_settle(hipSetDevice(h->uniq[u]), "hipSetDevice"))) return rc;
      if ((rc = sync_or_settle(hipStreamSynchronize(h->gstream[u]), "hipStreamSynchronize (gather)"))) return rc;
    }
    if (gather_ms) *gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (qilqr_solver *s : h->solvers)
      if (s->dev.profile) drain_events(s);
    for (int32_t r = 0; r < k; ++r)
      if ((rc = device_error(h->solvers[r]))) return fail(rc, "shard " + std::to_string(r) + " (device " + std::to_string(h->device[r]) + "): " + g_last_error);
    return QILQR_OK;
  } catch (...) {
    return fail(QILQR_ERR_INVALID_ARG, "qilqr_solve_batch_sharded_device: out of host memory");
  }
}

// In words: which arithmetic and which kernels a batch solve of B problems on this handle uses (the choice between the
// reference's own forms and the symmetric-weight kernels is made by whether Q and R are bit-exactly symmetric, and was silent).
int qilqr_describe(qilqr_solver *s, int32_t B, char *buf, size_t cap) {
  if (!s || !buf || cap == 0 || B <= 0) return fail(QILQR_ERR_INVALID_ARG, "bad argument");
  const long load_B = B;
  std::string t;
  const bool persistent = use_persistent(s, B);
  const BackwardKind kind = persistent ? BW_FOUR : backward_kind(s, load_B);
  if (kind == BW_ONE && !s->symmetric)
    t += "arithmetic: the reference's own forms (ilqr.hh:126-133: Eigen's diagonally pivoted LDL^T, V_x = Q_x - K^T Q_uu k, V_xx = Q_xx - K^T Q_uu K, "
         "not symmetrised)";
  else
    t += "arithmetic: symmetric-weight forms (Q == Q^T and R == R^T exactly: unpivoted LDL^T, V_x = Q_x + K^T Q_u, V_xx = Q_xx + Q_xu K on a symmetric "
         "accumulator; force_general = 1 selects the reference's own forms)";
  t += s->integrator == 1 ? "; Runge-Kutta step (extension)" : "; explicit Euler step (the reference's)";
  t += s->f32 ? "; mixed precision (fp32 storage and lane-local arithmetic, fp64 recursion and cost sums)" : "; fp64";
  t += "; backward: ";
  t += persistent ? "k_solve4 (one launch per solve)" : kind == BW_FUSED ? "k_backward4, fused matrix + gradient wavefronts" : kind == BW_FOUR ? "k_backward4, six wavefronts"
       : kind == BW_TWO ? "k_backward2" : (s->symmetric ? "k_backward<true>, one wavefront per trajectory" : "k_backward<false>, one wavefront per trajectory (general kernel)");
  if (kind == BW_FOUR || (kind == BW_FUSED && s->dev.force_general == 0 && load_B >= GFAC_MIN_LIVE))
    t += kind == BW_FOUR ? " (Q_uu factored by the gradient wavefront in launches with " + std::to_string(GFAC_MIN_LIVE) + " or more running trajectories, by the matrix wavefronts otherwise: the same bits)"
                         : " (six wavefronts, Q_uu factored by the gradient wavefront, while " + std::to_string(GFAC_MIN_LIVE) + " or more trajectories run: the same bits)";
  if (!persistent) {
    const int choice = s->dev.single_wave_rollout;
    t += "; rollout: ";
    t += (s->integrator == 1 || choice == 1) ? "k_rollout" : (choice == 3 || (choice == 0 && load_B <= R16_MAX_B)) ? "k_rollout16"
         : choice == 0 ? "k_rollout3 for a trajectory's first " + std::to_string(ROLLOUT16_FROM) + " rollouts, k_rollout16 from there on" : "k_rollout3";
    // (nothing of the handle is touched: the launch helpers take the batch and the record placement the call WOULD have)
    const bool tiled = records_tiled(s, B, persistent);
    const bool fused = fuse_backward_rollout(s, B, B, tiled) && s->dev.sync_every > 1;
    const int parts = s->dev.sync_every > 1 ? auto_parts(s, B) : 1;
    const bool compact = s->dev.compaction >= 0 && s->dev.sync_every > 1 &&
                         (s->dev.compaction == 1 || (!(fused && parts == 1) && kind != BW_ONE && tiled));
    // the round's launch form by the predicate run_solve uses: the combined launch only where the compaction does not work between its halves
    if (fused && parts == 1 && !compact)
      t += round_kernel_ok(s) ? "; round: one launch (k_round), " + std::to_string(rounds_per_launch(s)) + " rounds per launch, while no other batch solve of the process is in flight on the device"
                               : std::string("; round: k_backward_rollout + k_linearize");
    else if (compact && fuse_kinds(s, B, B, tiled))
      t += "; round: three launches while the compaction runs, then k_round (the mixed mode: k_backward_rollout + k_linearize) once the running trajectories fit " + std::to_string(std::min(parts, 3)) + " block(s) of four per CU";
    else if (long from = 0; compact && s->dev.compaction != 1 && late_tail_kinds(s, B, B, tiled, &from))
      t += "; round: three launches, then k_round (the same bits; the mixed mode: k_backward_rollout + k_linearize) once the running trajectories fit " + std::to_string(std::min(parts, 3)) + " block(s) of four per CU"
           + (from > 0 ? " and the rollouts are k_rollout16's (round " + std::to_string(from) + " on)" : std::string());
    else
      t += "; round: three launches";
    t += "; sub-batch streams: " + std::to_string(parts);
    t += compact ? "; compaction of the running trajectories: on (device-resident calls)" : "; compaction: off";
  }
  std::snprintf(buf, cap, "%s", t.c_str());
  return QILQR_OK;
}

// trajectories k_compact_move moved in the last batch solve of this handle (0: compaction was off, or nothing finished early)
int qilqr_compaction_moves(qilqr_solver *s, int64_t *moves) {
  if (!s || !moves) return fail(QILQR_ERR_INVALID_ARG, "null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  *moves = 0;
  for (long head : s->plan_heads) {
    int m = 0;
    HIP_TRY(hipMemcpy(&m, s->st.plan + head + 2, sizeof(int), hipMemcpyDeviceToHost));
    *moves += m;
  }
  return QILQR_OK;
}

#if defined(QILQR_STAMPS) || defined(QILQR_ROUND_STAMPS)
// diagnostic build only: per-trajectory section cycle sums of the last k_backward launch
int qilqr_debug_stamps(qilqr_solver *s, unsigned long long *out, int32_t B) {
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out, s->st.stamps, sizeof(unsigned long long) * 8 * B, hipMemcpyDeviceToHost));
  return QILQR_OK;
}
#endif

#ifdef QILQR_DIAG
// diagnostics build only: from now on the step wavefronts of k_rollout16 withhold the velocity hand-off of knot `knot` (-1: none
// again), on the device the solver is bound to -- the other wavefront's bounded spin then runs out and the block is abandoned
// 1: the late finishers of the copy-back under the tail go through the compact block and the host's scatter although the caller's arrays map
// (the fallback's parity test; A/B of the two forms)
int qilqr_debug_set_staged_late(int32_t on) {
  g_force_staged_late = on ? 1 : 0;
  return QILQR_OK;
}
int qilqr_debug_set_rollout_stall(qilqr_solver *s, int32_t knot) {
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_r16_stall_knot), &knot, sizeof(int)));
  return QILQR_OK;
}
// the same for k_backward4's barrier-free form: the loader withholds the tags of record `rec` (-1: none again)
int qilqr_debug_set_backward_stall(qilqr_solver *s, int32_t rec) {
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bw4_stall_rec), &rec, sizeof(int)));
  return QILQR_OK;
}
// diagnostic build only (make diag; profiles/microbench/beside.py): the rollout kernel of the
// state a qilqr_forward_sim call left behind, `reps` times on the solver's stream, alone (beside = 0) or while a second
// stream runs k_linearize launches back to back (beside = 1).  us[0]: microseconds per rollout launch; us[1]: per
// linearisation launch.  Timing only: the linearisation reads the candidate while it is being written.
int qilqr_debug_rollout_beside_linearize(qilqr_solver *s, int32_t B, int32_t n, int32_t reps, int32_t beside, float *us) {
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  if ((rc = ensure_parts(s, 1))) return rc;
  hipEvent_t e[4];
  for (auto &x : e) HIP_TRY(hipEventCreate(&x));
  HIP_TRY(hipStreamSynchronize(s->stream));
  hipStream_t main_stream = s->stream, side = s->part_stream[0];
  const int nlin = 4 * reps;
  if (beside) {
    s->stream = side;
    HIP_TRY(hipEventRecord(e[2], side));
    for (int k = 0; k < nlin; ++k)
      if ((rc = launch_linearize(s, B, n, 1, 0, -1))) break;
    HIP_TRY(hipEventRecord(e[3], side));
    s->stream = main_stream;
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(e[0], main_stream));
  for (int k = 0; k < reps; ++k)
    if ((rc = launch_rollout(s, B, n, 0))) return rc;
  HIP_TRY(hipEventRecord(e[1], main_stream));
  HIP_TRY(hipStreamSynchronize(main_stream));
  HIP_TRY(hipStreamSynchronize(side));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e[0], e[1]));
  us[0] = 1e3f * ms / reps;
  us[1] = 0.f;
  if (beside) {
    HIP_TRY(hipEventElapsedTime(&ms, e[2], e[3]));
    us[1] = 1e3f * ms / nlin;
  }
  for (auto &x : e) HIP_TRY(hipEventDestroy(x));
  return QILQR_OK;
}
#endif

}  // extern "C"
