// ilqr_kernels.h -- device kernels of the batched iLQR solver (gfx950 only).
//
// One outer "round" of the solver is three launches, back to back on one stream:
//   k_backward4 / k_backward   settles the previous candidate (cost sum, Armijo test, convergence tests:
//                ilqr.hh:61-84, 174-194), then the Riccati recursion on the fp64 matrix core
//                (ilqr.hh:97-147): four matrix wavefronts, a gradient and a loader wavefront per four trajectories
//                (k_backward4), or one wavefront per trajectory (k_backward)
//                (k_backward2 -- a matrix and a gradient wavefront per trajectory -- and the one-launch k_solve4 are in the
//                diagnostics build: -DQILQR_DIAG, `make diag`)
//   k_rollout3 / k_rollout     closed-loop forward simulation (ilqr.hh:149-172): pose wave + control
//                wave + loader wave per 64 trajectories, or one lane per trajectory in one wavefront
//   k_linearize  two lanes per knot: dynamics Jacobian blocks / cost differentials + knot cost of the
//                candidate trajectory (quadrotor_model.cc:33-49, cost.hh:36-61); also hands the
//                count of still-active trajectories to the host (pinned memory)
// (k_accept is the stand-alone acceptance step of the qilqr_line_search entry point.)
// Every trajectory carries its own outer-iteration counter, step size and state machine, so
// trajectories that are back-tracking and trajectories that already accepted a step advance
// in the same round; the host only reads one count of still-active trajectories, late.
#pragma once

// The kernels by family (round 5: one header per family; this file was 2 900 lines):
#include "kernels_common.h"        // workspace, parameters, state-machine stores
#include "linearize_kernels.h"     // k_linearize, k_begin, k_init
#include "backward_common.h"       // tile products, LDL^T, lane helpers
#include "backward1_kernel.h"      // k_backward: one wavefront per trajectory (the general kernel)
#include "backward2_kernel.h"      // k_backward2 (diagnostics build)
#include "backward4_kernel.h"      // k_backward4: four trajectories per block
#include "rollout_kernels.h"       // k_rollout, k_rollout3
#include "rollout16_kernel.h"      // k_rollout16
#include "round_kernels.h"         // k_backward_rollout, k_round
#include "bookkeeping_kernels.h"   // k_accept, k_gather, k_retile, compaction, debug capture


#ifdef QILQR_WITH_SOLVE4  // diagnostics build only (make diag): the one-launch solve, measured behind the rounds
#include "solve4.h"
#endif
