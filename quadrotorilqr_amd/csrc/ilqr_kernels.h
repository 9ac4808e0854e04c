// ilqr_kernels.h -- device kernels of the batched iLQR solver (gfx950 only).
//
// One outer "round" of the solver -- every running trajectory settles its pending candidate, runs the Riccati recursion, rolls out the next
// candidate and linearises it -- is, by how many trajectories a call has in flight:
//   up to 1024 (a block of four trajectories per CU)   ONE launch, k_round, and up to four rounds per launch: the three phases below as phases of
//                one kernel separated by block barriers (round_kernels.h; k_backward_rollout has the first two: the mixed-precision mode)
//   beyond       three launches back to back on the (sub-batch's) stream, with the compaction of the running trajectories (k_compact_plan,
//                k_compact_move) between the first two while it pays:
//   k_backward4 / k_backward   settles the previous candidate (cost sum, Armijo test, convergence tests:
//                ilqr.hh:61-84, 174-194), then the Riccati recursion on the fp64 matrix core
//                (ilqr.hh:97-147): per four trajectories four matrix-and-gradient wavefronts and a loader (the fused form), or four matrix
//                wavefronts, a gradient and a loader wavefront -- Q_uu factored by the gradient wavefront when the chip is saturated, by the
//                matrix wavefronts otherwise: one arithmetic in every form --, or one wavefront per trajectory (k_backward: general weights)
//   k_rollout16 / k_rollout3 / k_rollout   closed-loop forward simulation (ilqr.hh:149-172): sixteen lanes per trajectory (up to 4096
//                trajectories), pose wave + control wave + loader wave per 64 trajectories, or one lane per trajectory in one wavefront
//   k_linearize  two lanes per knot: dynamics Jacobian blocks / cost differentials + knot cost of the
//                candidate trajectory (quadrotor_model.cc:33-49, cost.hh:36-61); also hands the
//                count of still-active trajectories to the host (pinned memory)
// (k_accept is the stand-alone acceptance step of the qilqr_line_search entry point.  k_backward2 -- a matrix and a gradient wavefront per
// trajectory -- and the one-launch k_solve4 measured behind these and live in diag/, compiled into the diagnostics build only: -DQILQR_DIAG,
// `make diag`.)
// Every trajectory carries its own outer-iteration counter, step size and state machine, so
// trajectories that are back-tracking and trajectories that already accepted a step advance
// in the same round; the host only reads one count of still-active trajectories, late.
#pragma once

// The kernels by family (round 5: one header per family; this file was 2 900 lines):
#include "kernels_common.h"        // workspace, parameters, state-machine stores
#include "linearize_kernels.h"     // k_linearize, k_begin, k_init
#include "backward_common.h"       // tile products, LDL^T, lane helpers
#include "backward1_kernel.h"      // k_backward: one wavefront per trajectory (the general kernel)
#include "backward4_kernel.h"      // k_backward4: four trajectories per block
#include "rollout_kernels.h"       // k_rollout, k_rollout3
#include "rollout16_kernel.h"      // k_rollout16
#include "bookkeeping_kernels.h"   // k_accept, k_gather, k_retile, compaction, debug capture
#include "round_kernels.h"         // k_backward_rollout, k_round

// diagnostics build only (make diag): kernels that measured behind the product's, kept with their parity tests
#ifdef QILQR_WITH_BACKWARD2
#include "diag/backward2_kernel.h"  // k_backward2
#endif
#ifdef QILQR_WITH_SOLVE4
#include "diag/solve4.h"            // k_solve4: the one-launch solve
#endif
