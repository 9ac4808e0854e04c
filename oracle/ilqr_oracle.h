/*
 * ilqr_oracle.h -- CPU oracle for the SE(3) x R^6 quadrotor iLQR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quadrotorilqr_amd/,
 * src/, include/) may include, link or call this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * What it is: a dense, scalar, dependency-free C restatement of the reference
 * algorithm, following /root/reference/src/{ilqr.hh,cost.hh,quadrotor_model.cc}
 * operation by operation (each function cites the lines it follows).  The Lie
 * group arithmetic (manif @ ab560a3a, WORKSPACE:55-61) and the small dense
 * factorisations (Eigen 3.4.0, WORKSPACE:39-45) are third-party code that is NOT
 * in the reference tree; their published algorithms are restated here.
 *
 * Pinning status: the reference cannot be compiled in this image (Eigen/manif
 * absent).  The oracle is pinned against every known-answer and finite-difference
 * test the reference's own test-suite holds for this path (tests/test_oracle_*.py
 * list them with file:line), and cross-checked against scipy expm/logm.  For
 * converged trajectories on problems with coupled rotation+translation no
 * reference vector exists: for those, bit-level parity with manif/Eigen is
 * "parity unpinned" (see DESIGN.md).
 *
 * Conventions at this C interface
 *   state  x[13]  = [t(3) ; q = (w,x,y,z) ; v = (lin(3), ang(3))]
 *   tangent[12]   = [rho(3) ; theta(3) ; dv_lin(3) ; dv_ang(3)]   (manif order)
 *   knot   p[18]  = [time_s ; x(13) ; u(4)]   (= IDX in quadrotor_ilqr.py:19-37)
 *   matrices are row-major
 *   gains  g[52]  = [k(4) ; K(4x12) column-major]   (= the reference's
 *                   ControlUpdate{ff_update, feedback}, ilqr.hh:43-46)
 */
#ifndef ILQR_ORACLE_H
#define ILQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 12
#define ORC_NU 4
#define ORC_XS 13
#define ORC_PT 18
#define ORC_GAIN 52

/* solve() exit paths, numbered after the reference's return sites */
#define ORC_STATUS_CONVERGED_EXPECTED 0 /* ilqr.hh:66-68 */
#define ORC_STATUS_CONVERGED 1          /* ilqr.hh:82-84 */
#define ORC_STATUS_MAX_ITERS 2          /* ilqr.hh:86    */
#define ORC_STATUS_LINE_SEARCH_FAILED 3 /* ilqr.hh:191-193 (throws there) */

#define ORC_OK 0
#define ORC_ERR_BAD_INERTIA 1     /* quadrotor_model.cc:21-24 */
#define ORC_ERR_LENGTH_MISMATCH 2 /* cost.hh:39-40 (.at(i))   */
#define ORC_ERR_INVALID 3

typedef struct {
  double mass_kg;
  double inertia[9];
  double arm_length_m;
  double torque_to_thrust_ratio_m;
  double g_mpss;
} orc_model_params;

typedef struct {
  double step_update;
  double desired_reduction_frac;
  int ls_max_iters;
  double rtol;
  double atol;
  double max_iters; /* a double in the reference: ilqr_options.hh:14 */
  int populate_debug;
} orc_options;

typedef struct orc_solver orc_solver;

/* --- Lie group pieces (manif restatement), exposed for unit tests ------- */
void orc_so3_exp(const double th[3], double q_wxyz[4]);
void orc_so3_log(const double q_wxyz[4], double th[3]);
void orc_so3_ljac(const double th[3], double J[9]);
void orc_so3_ljacinv(const double th[3], double J[9]);
void orc_se3_exp(const double tau[6], double T[7]);     /* T = [t ; q wxyz] */
void orc_se3_log(const double T[7], double tau[6]);
void orc_se3_compose(const double A[7], const double B[7], double C[7]);
void orc_se3_inverse(const double A[7], double Ainv[7]);
void orc_se3_adj(const double T[7], double Ad[36]);
void orc_se3_rjac(const double tau[6], double J[36]);
void orc_se3_rjacinv(const double tau[6], double J[36]);
void orc_se3_ljacinv(const double tau[6], double J[36]);

/* --- model -------------------------------------------------------------- */
int orc_model_check(const orc_model_params *mp);
/* quadrotor_model.cc:65-122 */
int orc_continuous_dynamics(const orc_model_params *mp, const double x[13],
                            const double u[4], double xdot[12], double *Jx /*144|NULL*/,
                            double *Ju /*48|NULL*/);
/* quadrotor_model.cc:33-49 */
int orc_discrete_dynamics(const orc_model_params *mp, const double x[13],
                          const double u[4], double dt, double xnext[13],
                          double *Jx, double *Ju);
/* EXTENSION: the same with the integrator chosen -- 0 explicit Euler (= orc_discrete_dynamics), 1 the Runge-Kutta step
 * sketched in the comment at quadrotor_model.cc:51-63 (see discrete_dynamics_rk4 in ilqr_oracle.c) */
int orc_discrete_step(const orc_model_params *mp, int integrator, const double x[13], const double u[4],
                      double dt, double xnext[13], double *Jx, double *Ju);
/* quadrotor_model.cc:174-206 */
void orc_state_add(const double x[13], const double tangent[12], double out[13],
                   double *J_lhs, double *J_rhs);
/* quadrotor_model.cc:215-250 */
void orc_state_minus(const double lhs[13], const double rhs[13], double out[12],
                     double *J_lhs, double *J_rhs);
/* quadrotor_model.cc:266-276 */
void orc_euler_step(const double x[13], const double xdot[12], double dt,
                    double out[13], double *J_lhs, double *J_rhs);

/* --- cost (cost.hh:36-61) ----------------------------------------------- */
double orc_cost(const double Q[144], const double R[16], const double x[13],
                const double u[4], const double xd[13], const double ud[4],
                double *Cx, double *Cu, double *Cxx, double *Cuu, double *Cxu);

/* --- Eigen 4x4 pivoted LDLT solve (ilqr.hh:126-128) ---------------------- */
void orc_ldlt4_solve(const double A[16], const double *B, int nrhs, double *X);

/* --- solver (ilqr.hh) ---------------------------------------------------- */
int orc_solver_create(const orc_model_params *mp, const double Q[144],
                      const double R[16], const double *desired /*Nd x 18*/, int n_desired,
                      double dt, const orc_options *opt, orc_solver **out);
void orc_solver_destroy(orc_solver *s);

/* ilqr.hh:89-95 */
int orc_cost_trajectory(const orc_solver *s, const double *traj, int n, double *cost);
/* ilqr.hh:97-147; gains n x 52, terms = {QuTk, kTQuuk} */
int orc_backwards_pass(const orc_solver *s, const double *traj, int n, double *gains,
                       double terms[2]);
/* ilqr.hh:149-172 */
/* Extensions the reference does not have (Levenberg-Marquardt restarts, the counterpart of
 * qilqr_set_regularisation in include/quadrotor_ilqr.h): the backward pass with mu added to the diagonal
 * of Q_uu, and the switch that makes orc_solve / orc_solve_batch restart an exhausted line search. */
int orc_backwards_pass_reg(const orc_solver *s, const double *traj, int n, double mu, double *gains,
                           double terms[2]);
int orc_set_regularisation(orc_solver *s, double mu_init, double mu_factor, double mu_max);
/* EXTENSION: integrator of every pass of this solver, 0 (default, the reference) or 1 */
int orc_set_integrator(orc_solver *s, int integrator);
/* EXTENSION: form of the value recursion of every backward pass of this solver.  0 (default) = ilqr.hh:132-133 as the reference
 * writes it; 1 = the same update with K = -Q_uu^-1 Q_ux, k = -Q_uu^-1 Q_u substituted and V_xx symmetrised (V_x = Q_x + K^T Q_u,
 * V_xx = sym(Q_xx + Q_xu K), k^T Q_uu k = -Q_u^T k): algebraically the reference's, numerically stable at the 200- and 500-knot
 * horizons where the reference's own form is rounding noise.  The comparand of the full-size tests of BASELINE.json configs[2],
 * configs[4]; see the definition in ilqr_oracle.c. */
int orc_set_recursion(orc_solver *s, int mode);

int orc_forward_sim(const orc_solver *s, const double *traj, int n, const double *gains,
                    double alpha, double *out_traj);
/* ilqr.hh:174-194; returns ORC_STATUS_LINE_SEARCH_FAILED where the reference throws */
int orc_line_search(const orc_solver *s, const double *traj, int n, double cost,
                    const double *gains, const double terms[2], double *out_traj,
                    double *out_cost, double *out_step, int *out_trials);
/* ilqr.hh:53-87.  cost_hist (capacity cap) receives new_cost after every completed
 * forward pass (what ILQRDebug would hold); debug_trajs (cap x n x 18 or NULL) the
 * trajectories.  Counters: iters = completed outer iterations, n_bwd / n_fwd = passes. */
int orc_solve(const orc_solver *s, const double *init, int n, double *out_traj,
              double *out_cost, int *out_status, int *out_iters, int *out_n_bwd,
              int *out_n_fwd, double *cost_hist, double *debug_trajs, int cap,
              int *out_n_hist);

/* One comparison a solve's control flow depended on, with its margin: the distance of the compared quantity from the value at
 * which the comparison flips, as a cost difference over |cost| of the iteration (counts "must match
 * except where the deciding margin is < 1e-9 relative").  kind: ORC_DEC_EXPECTED = is_converged(cost, cost + dJ(1)) at the top of
 * iteration iter (ilqr.hh:66), ORC_DEC_ARMIJO = new_cost - cost < frac dJ(alpha) of trial `trial` (:186), ORC_DEC_CONVERGED =
 * is_converged(cost, new_cost) behind the accepted step (:82).  lhs / rhs: the two costs (convergence tests) or the two sides
 * of the Armijo inequality. */
#define ORC_DEC_EXPECTED 0
#define ORC_DEC_ARMIJO 1
#define ORC_DEC_CONVERGED 2
typedef struct {
  int kind, iter, trial, result;
  int n_bwd, n_fwd; /* backward passes / rollouts completed when the comparison is made (an Armijo test counts its own rollout) */
  double lhs, rhs, margin;
} orc_decision;
int orc_solve_decisions(const orc_solver *s, const double *init, int n, double *out_traj, double *out_cost,
                        int *out_status, int *out_iters, int *out_n_bwd, int *out_n_fwd, double *cost_hist, int cap,
                        int *out_n_hist, orc_decision *dec, int dec_cap, int *n_dec);

/* batch over independent problems, n_threads host threads (pthreads) drawing problems from an atomic counter.
 * init: B x n x 18; desired shared (from create).  Results do not depend on n_threads. */
int orc_solve_batch(const orc_solver *s, const double *init, int B, int n, double *out_traj,
                    double *out_cost, int *out_status, int *out_iters, int *out_n_bwd,
                    int *out_n_fwd, int n_threads);

/* "parity: ..." (the build every test compares with: no contraction) or "fast: <flags>" (timing only, bench.py's cpu_baseline) */
const char *orc_build_flavour(void);

#ifdef __cplusplus
}
#endif
#endif
