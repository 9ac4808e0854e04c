/*
 * quadrotor_ilqr.h -- C ABI of the MI355X-native batched iLQR solver for the
 * SE(3) x R^6 quadrotor.  This is the drop-in boundary for the one hot path of
 * nitishthatte/QuadrotorILQR: everything reachable from ILQR<QuadrotorModel>::solve
 * (reference src/ilqr.hh:53-87).  Plain pointers and sizes, no exceptions, no C++
 * or torch types.  The shared library is libquadrotor_ilqr.so (HIP, gfx950 only;
 * there is no CPU fallback: every entry point that computes fails with
 * QILQR_ERR_NO_DEVICE when no GPU is present).
 *
 * Conventions
 *   knot   p[18]  = [time_s, tx,ty,tz, qw,qx,qy,qz, v_lin(3), v_ang(3), u0..u3]
 *                   (= IDX of reference src/quadrotor_ilqr.py:19-37; quaternion in the
 *                   wire order w,x,y,z of src/trajectory.proto:27-30)
 *   trajectory    = n x 18 doubles, row-major; a batch is B x n x 18
 *   tangent order = [rho(3), theta(3), dv_lin(3), dv_ang(3)]  (quadrotor_model.hh:30-37)
 *   matrices      = row-major (Q is 12x12 in tangent order, R is 4x4)
 *   gains  g[52]  = [k(4) ; K(4x12) column-major]  = the reference's
 *                   ControlUpdate{ff_update, feedback} (ilqr.hh:43-46)
 */
#ifndef QUADROTOR_ILQR_H
#define QUADROTOR_ILQR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QILQR_KNOT 18
#define QILQR_GAIN 52

/* return codes */
#define QILQR_OK 0
#define QILQR_ERR_BAD_INERTIA 1     /* "Inertia matrix is not positive definite!" quadrotor_model.cc:21-24 */
#define QILQR_ERR_LENGTH_MISMATCH 2 /* initial trajectory longer than desired: cost.hh:39-40 (.at) */
#define QILQR_ERR_INVALID_ARG 3
#define QILQR_ERR_BAD_QUATERNION 4  /* manif's SO3 constructor check, | |q| - 1 | > 1e-10 */
#define QILQR_ERR_NO_DEVICE 5
#define QILQR_ERR_HIP 6
#define QILQR_ERR_LINE_SEARCH 7     /* single solve only: ilqr.hh:191-193 throws */

/* per-problem exit path of solve(), numbered after the reference's return sites */
#define QILQR_STATUS_CONVERGED_EXPECTED 0 /* ilqr.hh:66-68 */
#define QILQR_STATUS_CONVERGED 1          /* ilqr.hh:82-84 */
#define QILQR_STATUS_MAX_ITERS 2          /* ilqr.hh:86    */
#define QILQR_STATUS_LINE_SEARCH_FAILED 3 /* ilqr.hh:191-193 */

/* QuadrotorModel constructor arguments: quadrotor_model.hh:7-9, binding.cc:20-23 */
typedef struct {
  double mass_kg;
  double inertia[9];
  double arm_length_m;
  double torque_to_thrust_ratio_m;
  double g_mpss;
} qilqr_model;

/* ILQROptions: ilqr_options.hh:4-22 / ilqr_options.proto:5-21 */
typedef struct {
  double step_update;            /* LineSearchParams */
  double desired_reduction_frac;
  int32_t ls_max_iters;
  double rtol;                   /* ConvergenceCriteria */
  double atol;
  double max_iters;              /* a double in the reference */
  int32_t populate_debug;
} qilqr_options;

/* device-side configuration (no counterpart in the reference, which is CPU only) */
typedef struct {
  int32_t device;   /* HIP device ordinal */
  int32_t profile;  /* start/stop HIP events attached to kernel dispatches.  Low byte: 0 off; 1 k_backward and
                       k_rollout; 2 every kernel; 3 k_backward only; 4 k_rollout only.  Second byte: sampling
                       stride s (0 or 1: every selected launch; s > 1: every s-th launch of a kind is timed, the
                       averages of qilqr_profile_get are over the timed launches).  Bit 16 (0x10000): roctx ranges -- the host thread
                       marks a batch solve, every round it enqueues and, for a batch on sub-batch streams, every part's share of a round
                       ("qilqr round 17 part 2"), for `rocprofv3 --marker-trace --kernel-trace` (libroctx64 is bound at first use; absent,
                       no ranges) */
  int32_t sync_every; /* 1: the host waits for every round's count of active trajectories; k > 1 (default 2 when no
                         configuration is given): it reads the count k rounds late, i.e. keeps the stream k rounds
                         ahead of the device (k <= 6).  The results do not depend on it. */
  int32_t force_general; /* backward kernel.  0: by the weights and the batch (symmetric Q, R: k_backward4 in its fused form --
                            four wavefronts that each carry the matrix and the gradient recursion of a trajectory, plus a
                            loader wavefront, per four trajectories -- up to 4096 trajectories; in its six-wavefront form --
                            four matrix wavefronts, one gradient and one loader wavefront -- beyond (until ABI version 6 one wavefront
                            per trajectory took over above 8192: with the running trajectories compacted, see `compaction`, the blocks of
                            four are ahead at every size); non-symmetric Q or R: the general kernel); 1: the general kernel even when
                            Q, R are symmetric; 2: the one-wavefront kernel for symmetric weights (k_backward<true>);
                            3: k_backward2 (diagnostics build only); 4: k_backward4, six wavefronts (Q_uu factored by the gradient
                            wavefront in launches with 3072 or more running trajectories, by the matrix wavefronts otherwise; 7 / 8:
                            the one / the other at every launch -- 8 where the round is the combined launch k_round: its backward phase in
                            the six-wavefront form in every launch, which 0 takes for the launches in which at most two trajectories per
                            block still run); 5: k_backward4, fused
                            (its wavefronts meet through tagged LDS slots, no block barrier in the knot loop: what 0
                            selects up to 4096 trajectories; forced, it is used at every size); 6: retired in round 4
                            (the fused form with a block barrier per knot: refused by name).  When the round's kernels are the fused k_backward4 and k_rollout16 and
                            every block of four trajectories has a CU to itself (B <= 4 x the device's CUs: 1024 on
                            MI355X), the two are ONE launch (k_backward_rollout: the block's backward pass, a block
                            barrier, the rollout of its own four trajectories; same arithmetic, same bits) -- in fp64 together with
                            the linearisation of the block's candidates, four rounds to a launch (k_round: same bits again).
                            Since ABI version 7 / round 6 every form of k_backward4 performs ONE arithmetic (H accumulated in one
                            order, the gradient's sums in one order, Q_uu the same bits whoever factors it): a backward pass does not
                            depend on the batch size, and 0 also takes the six-wavefront form for the launches of a batch of up to 4096
                            in which 3072 or more trajectories still run.
                            WHICH ARITHMETIC A CALLER GETS.  The general kernel evaluates ilqr.hh:118-140 in the
                            reference's own forms: Q_uu factored by Eigen's diagonally pivoted LDL^T (largest |d_ii| of
                            the trailing block, first on ties), V_x = Q_x - K^T Q_uu k, V_xx = Q_xx - K^T Q_uu K, not
                            symmetrised -- force_general = 1 evaluates THE REFERENCE'S FORMULAS (the same products and sums; the
                            grouping of M^T V M alternates with the knot's parity -- (M^T V) M at odd knots, M^T (V M) at even ones --
                            because the accumulator tile is used transposed every other knot: not one fixed evaluation order of
                            ilqr.hh:118-124, equal to it to rounding), including the reference's loss of accuracy
                            beyond about 150 knots (the unsymmetrised recursion amplifies rounding asymmetry until the
                            gains are noise -- in the reference, the oracle and this kernel alike, of different magnitudes:
                            DESIGN.md section 4).  The symmetric-weight kernels (selected silently
                            whenever Q == Q^T and R == R^T exactly, i.e. for every weight the reference's demo and tests
                            use) DELIBERATELY DIFFER: unpivoted LDL^T (same result in exact arithmetic for positive
                            definite Q_uu; loses eps / p for an indefinite Q_uu with a tiny leading entry p) and the
                            equivalent forms V_x = Q_x + K^T Q_u, V_xx = Q_xx + Q_xu K on a symmetric accumulator, which
                            agree with the reference to rounding up to about 150 knots and stay bounded and convergent
                            beyond, where the reference's results are rounding noise. */
  int32_t single_wave_rollout; /* rollout kernel: 0 (default) = by the batch: sixteen lanes per trajectory, four
                                  trajectories per block (k_rollout16) up to 4096 trajectories; beyond, a lane per trajectory
                                  in three cooperating wavefronts (k_rollout3) for a trajectory's first 16 rollouts and k_rollout16
                                  from its 17th on (round 6: by then a fifth of a batch still runs and the kernels are lone dependent
                                  chains, where sixteen lanes per trajectory are a third faster; a running trajectory rolls out once per
                                  round, so the ordinal of a rollout is the round in EVERY call -- the choice is a property of the
                                  problem and of which side of 4096 its call is on, never of the other problems in the batch);
                                  1 = k_rollout (a lane per trajectory, one wavefront: the Runge-Kutta extension's kernel);
                                  2 = k_rollout3; 3 = k_rollout16 */
  int32_t precision; /* 0: fp64 everywhere (reference parity).  1: mixed: trajectories, gains and knot records
                        stored in fp32, rollout and linearisation computed in fp32, Riccati recursion on the fp64
                        matrix core, cost sums / Armijo / convergence tests in fp64 (BASELINE.json configs[2]) */
  int32_t streams; /* batch solves with sync_every > 1: number of contiguous sub-batches that run their rounds on
                      their own HIP streams (their kernels are bound by different resources and overlap);
                      0 = automatic (1 below 4096 trajectories, 2 at 4096, beyond 4 when the process runs with
                      GPU_MAX_HW_QUEUES >= 8 and 2 otherwise: see auto_parts in ilqr_capi.hip), at most 8 */
  int32_t persistent; /* the solve as ONE launch (k_solve4: blocks of eight wavefronts own four trajectories each from the
                         first linearisation to the exit status, no rounds, no host in the loop; symmetric weights only):
                         0 = the rounds of three launches at every batch size (by measurement they are level or ahead at every size
                         but one, DESIGN.md section 4: the library never selects the one-launch solve by itself), 1 = always k_solve4,
                         2 = never (the same as 0 today) */
  int32_t compaction; /* (ABI version 6) device-resident batch solves with sync_every > 1: between a round's backward pass and its
                         rollout the trajectories still running are moved into a dense prefix of the workspace, the finished
                         ones they replace leaving for the caller's result arrays at once -- a batch takes as many rounds as its
                         slowest problem and the kernels hand out work in groups of 4 and 64 trajectories that cost the same
                         with one running as with all.  Results are bit-identical with and without.  0 = automatic (whenever the
                         round's backward pass is a k_backward4 launch of its own, i.e. symmetric weights and more than 1024
                         trajectories, while more than 512 of a sub-batch are running, and until the running ones fit the combined launch
                         -- k_backward_rollout, then k_round -- which the rounds then change over to: at once in a call of up to 4096
                         trajectories, in a larger one from the round in which its rollouts are k_rollout16's anyway, see
                         single_wave_rollout), 1 = at every size and count the call allows
                         (not with populate_debug's cost history, per-problem desired trajectories, or the copy-back under the
                         tail of qilqr_solve_batch), -1 = never */
  int32_t round_launch; /* (ABI version 7; until then the environment variables QILQR_FUSE_BACKWARD_ROLLOUT / QILQR_ROUND_KERNEL) how a round of
                           up to 4 x CUs trajectories is launched when its kernels are the fused k_backward4 and k_rollout16: 0 = automatic
                           (ONE launch: k_round, fp64 -- backward pass, rollout and linearisation of the candidates -- or k_backward_rollout +
                           k_linearize in the mixed mode), 1 = three launches (k_backward4, k_rollout16, k_linearize), 2 = two launches
                           (k_backward_rollout + k_linearize).  The same bits in every form: A/B measurements and the bit-identity tests */
  int32_t rounds_per_launch; /* (ABI version 7; was QILQR_ROUNDS_PER_LAUNCH) rounds in one k_round launch where a launch may hold several:
                                0 = automatic (4), or 1, 2, 4 */
  int32_t fuse_in_flight; /* (ABI version 7; was QILQR_FUSE_IN_FLIGHT) 1 = keep the combined launches although other batch solves of the
                             process are in flight on the device (0: a solve that is not alone launches the kernels apart) */
  int32_t dense_weights; /* (ABI version 7; was QILQR_NO_DIAG_Q) 1 = a diagonal Q is multiplied as a dense matrix instead of scaling rows
                            (the same bits: tests/test_gpu_parity.py::test_diagonal_weights_path_gives_the_same_bits) */
} qilqr_device_config;
/* The structure only ever grows at its end.  qilqr_create_sized / qilqr_sharded_create_sized take the size the CALLER was compiled with
 * (fields beyond it keep their defaults: 0, sync_every 2), so a caller built against an older header runs against a newer library;
 * with this header `qilqr_create(...)`, `qilqr_sharded_create(...)` and `qilqr_sharded_create_mask(...)` in source code ARE the sized calls
 * (macros below).  The exported symbols of those three names remain for binaries built before ABI version 7 -- and for callers that bind
 * symbols by name (dlsym, ctypes, function pointers: bind the *_sized names instead): they read ONLY the eight fields of ABI version 5
 * (32 bytes) and IGNORE every field behind them, whatever the caller's structure holds -- `compaction` of ABI version 6 (36 bytes: a
 * version-6 binary's value is not honoured through the raw symbols) and the four switches of version 7 keep their defaults there. */
#define QILQR_DEVICE_CONFIG_BYTES_ABI5 32

/* One arithmetic at every batch size: k_rollout16 is forced, so that a problem's result does not depend on the size of the batch it is
 * solved in, on sharding, or on sub-batch streams (the reference is trivially batching-independent: it solves one problem per call).
 * Since round 6 the backward pass needs no pinning (every form of k_backward4 performs the same arithmetic; the macro no longer forces
 * the fused form, which cost throughput above 4096 trajectories per call); the sixteen-lane rollout still costs some there. */
#define QILQR_PIN_ARITHMETIC(cfg) do { (cfg).single_wave_rollout = 3; } while (0)

/* A handle owns its device workspace and stream: use it from one thread at a time (different handles are
 * independent; the reference's ILQR object is const and re-entrant). */
typedef struct qilqr_solver qilqr_solver;

/* per-kernel device time accumulated since the last reset: *_ms and *_launches cover the launches that carried
 * events (kernels not selected by `profile`, and launches skipped by the sampling stride, do not), *_seen counts
 * every launch of the kind while profiling was on */
typedef struct {
  double backward_ms;  int32_t backward_launches;
  double rollout_ms;   int32_t rollout_launches;
  double linearize_ms; int32_t linearize_launches;
  double other_ms;     int32_t other_launches;
  int32_t backward_seen, rollout_seen, linearize_seen, other_seen;
  double solve_ms;     int32_t solve_launches;  /* persistent solves (k_solve4: one launch per batch solve) */
  int32_t solve_seen;
} qilqr_profile;

/* Replaces src::init, quadrotor_ilqr_binding.cc:20-32 (QuadrotorModel ctor + CostFunction +
 * ILQR ctor).  `desired` is n_desired x 18 (time column ignored).  Everything is copied. */
int qilqr_create(const qilqr_model *model, const double *Q, const double *R,
                 const double *desired, int32_t n_desired, double dt_s,
                 const qilqr_options *options, const qilqr_device_config *dev,
                 qilqr_solver **out);
int qilqr_create_sized(const qilqr_model *model, const double *Q, const double *R,
                       const double *desired, int32_t n_desired, double dt_s,
                       const qilqr_options *options, const qilqr_device_config *dev, size_t dev_bytes,
                       qilqr_solver **out);
void qilqr_destroy(qilqr_solver *s);

/* text of the last error on the calling thread */
const char *qilqr_last_error(void);

/* Replaces src::solve, quadrotor_ilqr_binding.cc:34-41 -> ILQR::solve, ilqr.hh:53-87.
 * One problem.  debug_cost[debug_cap] / debug_trajs[debug_cap x n x 18] receive, when
 * options.populate_debug, one entry per completed forward pass (ilqr.hh:78-80); n_debug the
 * count.  Returns QILQR_ERR_LINE_SEARCH where the reference throws (outputs untouched). */
int qilqr_solve(qilqr_solver *s, const double *init, int32_t n, double *out_traj,
                double *out_cost, int32_t *out_status, int32_t *out_iters,
                double *debug_cost, double *debug_trajs, int32_t debug_cap, int32_t *n_debug);

/* B independent problems sharing model, cost weights, dt and options.  Host buffers.
 * init B x n x 18.  desired_batch: NULL (use the desired trajectory given at create) or
 * B x n x 18 per-problem desired trajectories.  Any output pointer may be NULL.
 * Line-search exhaustion is reported per problem in out_status, not as an error. */
int qilqr_solve_batch(qilqr_solver *s, const double *init, const double *desired_batch,
                      int32_t B, int32_t n, double *out_traj, double *out_cost,
                      int32_t *out_status, int32_t *out_iters, int32_t *out_n_bwd,
                      int32_t *out_n_fwd);

/* Same, with every buffer already resident in device memory (HBM) of the solver's device: plain contiguous
 * B x n x 18 doubles / B doubles / B int32, caller-owned, outputs at least that large (nothing is checked on the
 * device side; any output may be NULL).  Stream ordering: the solve runs on the solver's OWN stream (qilqr_stream,
 * created non-blocking: it is not ordered with the null stream or with any other stream).  Inputs written by
 * asynchronous work on another stream must be ordered first: record an event there and pass it to
 * qilqr_stream_wait_event (or synchronise that stream).  The call returns after the solver's stream has drained,
 * so the outputs may be read from any stream afterwards. */
int qilqr_solve_batch_device(qilqr_solver *s, const double *d_init, const double *d_desired_batch,
                             int32_t B, int32_t n, double *d_out_traj, double *d_out_cost,
                             int32_t *d_out_status, int32_t *d_out_iters, int32_t *d_out_n_bwd,
                             int32_t *d_out_n_fwd);

/* The passes the reference's tests call directly (ilqr_test.cc:102-190), batched, host buffers. */
/* ILQR::cost_trajectory, ilqr.hh:89-95 */
int qilqr_cost_trajectory(qilqr_solver *s, const double *traj, int32_t B, int32_t n, double *cost);
/* ILQR::backwards_pass, ilqr.hh:97-147: gains B x n x 52, terms B x 2 = {QuTk, kTQuuk} */
int qilqr_backwards_pass(qilqr_solver *s, const double *traj, int32_t B, int32_t n, double *gains,
                         double *terms);
/* ILQR::forward_sim, ilqr.hh:149-172: alpha[B] */
int qilqr_forward_sim(qilqr_solver *s, const double *traj, const double *gains,
                      const double *alpha, int32_t B, int32_t n, double *out_traj);
/* ILQR::line_search, ilqr.hh:174-194: cost[B], terms B x 2 -> out_traj, out_cost[B], out_step[B],
 * out_status[B] (0 accepted, QILQR_STATUS_LINE_SEARCH_FAILED where the reference throws) */
int qilqr_line_search(qilqr_solver *s, const double *traj, const double *cost, const double *gains,
                      const double *terms, int32_t B, int32_t n, double *out_traj,
                      double *out_cost, double *out_step, int32_t *out_status);

/* Cost history of the last batch solve (options.populate_debug = 1): hist is B x cap, row b holds
 * new_cost after each completed forward pass of problem b (what ILQRDebug.cost would hold, ilqr.hh:78-80;
 * trajectories are only captured by the single-problem qilqr_solve), unused entries are NaN.
 * cap = the largest number of entries any problem can have (= min(ceil(max_iters), 1e6): the loop of ilqr.hh:58
 * runs while i < max_iters, a double); returned in *out_cap
 * when hist is NULL. */
int qilqr_cost_history(qilqr_solver *s, int32_t B, double *hist, int32_t cap, int32_t *out_cap);

/* profiling (HIP events on the solver's stream) */
int qilqr_profile_reset(qilqr_solver *s);
int qilqr_profile_get(qilqr_solver *s, qilqr_profile *out);
/* change qilqr_device_config.profile of a live solver (drains the stream, resets the accumulated times) */
int qilqr_profile_mode(qilqr_solver *s, int32_t mode);

/* Levenberg-Marquardt restarts -- an EXTENSION the reference does not have (SURVEY.md section 8f row 4;
 * BASELINE.json configs[4] "line-search/regularisation restarts"); off by default and when mu_init == 0,
 * and then every result is the reference's.  Where ILQR::line_search would throw after ls_max_iters trials
 * (ilqr.hh:191-193) the problem instead keeps its iterate, sets mu = mu_init (then mu *= mu_factor on each
 * further exhaustion), repeats ILQR::backwards_pass with Q_uu + mu 1 in place of Q_uu in every formula of
 * ilqr.hh:126-140, and searches again from a full step.  An accepted step divides mu by mu_factor (below
 * mu_init it returns to 0).  Past mu_max the problem ends with QILQR_STATUS_LINE_SEARCH_FAILED as before.
 * Restarts are not iterations; they show in out_n_bwd.  Requires mu_factor > 1, mu_max >= mu_init. */
int qilqr_set_regularisation(qilqr_solver *s, double mu_init, double mu_factor, double mu_max);

/* Runge-Kutta integration of the dynamics -- an EXTENSION: the four-stage step that
 * quadrotor_model.cc:51-63 sketches in a comment and the reference never executes,
 *     k_0 = f(x, u), k_i = f(x (+) h_i k_{i-1}, u) with h = {0, dt/2, dt/2, dt};  x_next = x (+) dt (k_0 + 2 k_1 + 2 k_2 + k_3) / 6,
 * in place of the explicit Euler step of quadrotor_model.cc:33-49 in every pass (forward simulation, and the Jacobians
 * J_x, J_u of the backward pass by the chain rule through the stages).  integrator = 0 (default): the reference's step, and
 * then every result is the reference's; 1: the extension (fp64 solvers only).  Stated in the oracle from the reference's
 * own primitives; measured order of accuracy on SE(3): two, against Euler's
 * one (tests/test_oracle_rk4.py explains why not four).  The extension runs on the general kernels -- a lane per
 * trajectory rollout, the one-wavefront backward pass over dense Jacobian records -- not on the tuned Euler path. */
int qilqr_set_integrator(qilqr_solver *s, int32_t integrator);

/* device the solver is bound to, and the HIP stream it launches on (hipStream_t as void*) */
int qilqr_device(const qilqr_solver *s);
void *qilqr_stream(const qilqr_solver *s);
/* make the solver's stream wait (on the device) for a hipEvent_t recorded on another stream */
int qilqr_stream_wait_event(qilqr_solver *s, void *hip_event);

/* Pinned host memory for the buffers handed to the host-buffer entry points (qilqr_solve_batch copies with plain
 * hipMemcpy: direct DMA from / to pinned memory, HIP's chunked staging for pageable memory).  NULL on failure. */
void *qilqr_host_alloc(size_t bytes);
void qilqr_host_free(void *p);

/* ---- one batch over several devices, in ONE process (BASELINE.json configs[3]: independent problems, contiguous shards,
 * no exchange between them -- the reference has no counterpart: its ILQR object solves one problem on one core).
 * A sharded handle owns one qilqr_solver per entry of `devices` (an ordinal may repeat: two shards then overlap on that
 * device through two handles and two streams).  qilqr_solve_batch_sharded cuts the B problems into n_devices contiguous
 * shards in the order of `devices` -- B / n_devices each, the first B % n_devices one more (qilqr_shard_range; the rule of
 * quadrotorilqr_amd/sharding.py for the one-process-per-GPU deployment) -- and solves shard r on devices[r] from a host
 * thread of its own: its input slice goes to the device, its results come back into the caller's arrays at the shard's
 * offset (the "gather" is the copy back itself: ragged shards need no padding), the call returns when every shard has.
 * Arguments and results are those of qilqr_solve_batch.  Problem by problem they are bit-identical to a single-device solve
 * of the same batch WHEN SHARD AND WHOLE BATCH TAKE THE SAME ROLLOUT KERNEL.  The backward pass, the linearisation, the cost sums and
 * every decision are one arithmetic at every batch size (since round 6: tests/test_gpu_parity.py::
 * test_backward_pass_bits_do_not_depend_on_the_batch_size); what remains is the rollout: with single_wave_rollout = 0 a call with up to
 * 4096 trajectories in flight on its device takes k_rollout16 (sixteen lanes per trajectory) for every rollout and a larger one k_rollout3
 * (a lane per trajectory) for a trajectory's first 16 rollouts and k_rollout16 from the 17th on -- a rule in the call's side of 4096 and
 * the rollout's ordinal, which is the round number in every call, never in what else the batch holds --, the two kernels evaluate the
 * same formulas in different orders, and the same problem differs by about 1e-10 relative in its trajectory between, say, a batch of
 * 8192 and its eight shards of 1024 (a batch of 65536 and its shards of 8192 are on one side: the same bits) -- the exit path of a
 * problem that sits within rounding of a convergence threshold can differ with them.  Forcing one rollout kernel (single_wave_rollout = 2 or 3; QILQR_PIN_ARITHMETIC below) makes a problem's bits
 * independent of how the caller batches or shards it.  A shard that fails makes the call return its error (the lowest failing shard's; text through
 * qilqr_last_error, prefixed with the shard and device); the other shards still complete. */
typedef struct qilqr_sharded qilqr_sharded;
/* dev: as for qilqr_create, its `device` field is ignored (NULL = defaults) */
int qilqr_sharded_create(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                         int32_t n_desired, double dt_s, const qilqr_options *options, const qilqr_device_config *dev,
                         const int32_t *devices, int32_t n_devices, qilqr_sharded **out);
int qilqr_sharded_create_sized(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                               int32_t n_desired, double dt_s, const qilqr_options *options, const qilqr_device_config *dev,
                               size_t dev_bytes, const int32_t *devices, int32_t n_devices, qilqr_sharded **out);
/* the same with the devices given as a bit mask (bit d = HIP device d), lowest ordinal first */
int qilqr_sharded_create_mask_sized(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                                    int32_t n_desired, double dt_s, const qilqr_options *options,
                                    const qilqr_device_config *dev, size_t dev_bytes, uint64_t device_mask, qilqr_sharded **out);
int qilqr_sharded_create_mask(const qilqr_model *model, const double *Q, const double *R, const double *desired,
                              int32_t n_desired, double dt_s, const qilqr_options *options,
                              const qilqr_device_config *dev, uint64_t device_mask, qilqr_sharded **out);
void qilqr_sharded_destroy(qilqr_sharded *h);
int32_t qilqr_sharded_count(const qilqr_sharded *h);
/* the solver of shard r (to set regularisation, read profiles or cost histories shard by shard); NULL if r is out of range */
qilqr_solver *qilqr_sharded_solver(qilqr_sharded *h, int32_t r);
/* problems [*begin, *begin + *count) of a batch of B belong to shard r of n_shards */
int qilqr_shard_range(int32_t B, int32_t n_shards, int32_t r, int32_t *begin, int32_t *count);
int qilqr_solve_batch_sharded(qilqr_sharded *h, const double *init, const double *desired_batch, int32_t B, int32_t n,
                              double *out_traj, double *out_cost, int32_t *out_status, int32_t *out_iters,
                              int32_t *out_n_bwd, int32_t *out_n_fwd);

/* The same batch solve with the results gathered in ONE device's memory (BASELINE.json configs[3]: "sharded ... with RCCL
 * gather over xGMI"; the C counterpart of quadrotorilqr_amd/sharding.gather_to_root for a host that drives every GPU from
 * one process).  Inputs are host arrays as for qilqr_solve_batch_sharded; d_out_* are device arrays on the device of shard
 * `root` (devices[root]), B x n x 18 doubles / B doubles / B int32, any of them may be NULL.  Every shard's rows travel from
 * its solver's staging buffers straight into their place in the root's arrays -- ragged shards, no padding, no second
 * copy -- as soon as that shard has finished, by the handle's transport:
 *   QILQR_TRANSPORT_RCCL       ncclSend on the shard's device / ncclRecv on the root's, ONE GROUP PER SHARD, issued by that
 *                              shard's own host thread when its solve has finished (a communicator executes in issue order:
 *                              the groups reach the root's in the order the shards finish, so a shard's rows travel while
 *                              slower shards still solve), over one communicator per distinct
 *                              device (ncclCommInitAll: all in this process); librccl.so.1 is loaded when the first
 *                              communicator is needed.  Exercised with ONE rank so far (every shard on the one GPU of the
 *                              test box: self send / receive); the multi-rank path has not run on hardware -- its schedule
 *                              (ranks, offsets, counts) is checked on the CPU through qilqr_gather_schedule
 *   QILQR_TRANSPORT_PEER_COPY  hipMemcpyPeerAsync
 *   QILQR_TRANSPORT_AUTO       (default) RCCL when the shards sit on more than one device, device copies when they all share
 *                              one; falls back to peer copies if RCCL cannot be loaded or initialised
 * qilqr_sharded_transport says in words which one a handle uses (and why, after a fallback); forcing _RCCL fails instead of
 * falling back.  gather_ms (may be NULL): the exposed part of the gather -- from the moment the slowest shard's solve has
 * finished to the moment the root holds every row.  Results are, problem by problem, those of qilqr_solve_batch. */
#define QILQR_TRANSPORT_AUTO 0
#define QILQR_TRANSPORT_RCCL 1
#define QILQR_TRANSPORT_PEER_COPY 2
int qilqr_sharded_set_transport(qilqr_sharded *h, int32_t transport);
const char *qilqr_sharded_transport(qilqr_sharded *h);
int qilqr_solve_batch_sharded_device(qilqr_sharded *h, const double *init, const double *desired_batch, int32_t B, int32_t n,
                                     int32_t root, double *d_out_traj, double *d_out_cost, int32_t *d_out_status,
                                     int32_t *d_out_iters, int32_t *d_out_n_bwd, int32_t *d_out_n_fwd, double *gather_ms);

/* The transfers qilqr_solve_batch_sharded_device issues for a batch of B problems of n knots over n_shards shards on `devices`
 * (HIP ordinals; equal ordinals share a communicator rank, ranks numbered in order of first appearance), gathered on shard
 * `root`'s device -- computed, not issued: no device is touched, so a host without eight GPUs can check the schedule of eight.
 * arrays: bit mask of the outputs asked for (1 traj, 2 cost, 4 status, 8 iters, 16 n_bwd, 32 n_fwd).  out receives 7 int64 per
 * transfer, {shard, array (0 traj .. 5 n_fwd), src_rank, dst_rank, src_off, dst_off, count} -- offsets and counts in elements
 * of the array's type; src_off into the shard's staging buffer (the four int32 arrays sit one behind the other there), dst_off
 * into the root's array -- shard by shard in the order a shard enqueues them (at run time each shard's transfers are ONE
 * ncclGroup of send / receive pairs, or peer copies, enqueued by the shard's own host thread behind its solve).  Returns the
 * number of transfers (at most `cap` are written; out may be NULL), -1 on bad arguments. */
int qilqr_gather_schedule(int32_t B, int32_t n, const int32_t *devices, int32_t n_shards, int32_t root, uint32_t arrays,
                          int64_t *out, int32_t cap);

/* trajectories moved by the compaction (qilqr_device_config.compaction) in the last batch solve of this handle; 0 when it was
 * off for that call.  Waits for the handle's stream. */
int qilqr_compaction_moves(qilqr_solver *s, int64_t *moves);

/* In words, which arithmetic and which kernels a batch solve of B problems on this handle uses: the reference's own forms or the
 * symmetric-weight forms (the choice is made by whether Q and R are bit-exactly symmetric and by force_general: see
 * qilqr_device_config.force_general), the integrator, the precision, the backward and rollout kernels, how a round is launched,
 * sub-batch streams, compaction.  buf receives a NUL-terminated text of at most cap - 1 characters. */
int qilqr_describe(qilqr_solver *s, int32_t B, char *buf, size_t cap);

/* ABI version of this header: 7 (qilqr_device_config grew by round_launch, rounds_per_launch, fuse_in_flight, dense_weights -- the
 * switches that were environment variables -- and the *_sized entry points carry the caller's structure size; version 6 added
 * `compaction`) */
#define QILQR_ABI_VERSION 7
int qilqr_abi_version(void);

#ifdef __cplusplus
}
#endif

/* In source code compiled against this header the create calls pass the size of the structure they were compiled with. */
#ifndef QILQR_NO_SIZED_MACROS
#define qilqr_create(model, Q, R, desired, n_desired, dt_s, options, dev, out) \
  qilqr_create_sized(model, Q, R, desired, n_desired, dt_s, options, dev, sizeof(qilqr_device_config), out)
#define qilqr_sharded_create(model, Q, R, desired, n_desired, dt_s, options, dev, devices, n_devices, out) \
  qilqr_sharded_create_sized(model, Q, R, desired, n_desired, dt_s, options, dev, sizeof(qilqr_device_config), devices, n_devices, out)
#define qilqr_sharded_create_mask(model, Q, R, desired, n_desired, dt_s, options, dev, device_mask, out) \
  qilqr_sharded_create_mask_sized(model, Q, R, desired, n_desired, dt_s, options, dev, sizeof(qilqr_device_config), device_mask, out)
#endif
#endif
