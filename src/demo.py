"""The build's counterpart of the reference's demo driver (behaviour of src/quadrotor_ilqr.py:256-306,
without the matplotlib / STL visualisation, which is out of scope): builds the box-climb desired
trajectory and the default options as protobuf messages, calls the binding exactly as the reference
does, and returns the optimised trajectory and cost history as arrays.

    python -m src.demo [--horizon_s 4.0]
"""
import argparse

import numpy as np

import src.ilqr_options_pb2 as opts
import src.trajectory_pb2 as traj
from quadrotorilqr_amd import problems as pb
from src.quadrotor_ilqr_binding import QuadrotorILQR, _decode_trajectory


def trajectory_message(arr):
    """(n, 18) array in IDX order (quadrotor_ilqr.py:19-37) -> QuadrotorTrajectory"""
    pts = []
    for k in np.asarray(arr, dtype=float):
        pts.append(traj.QuadrotorTrajectoryPoint(
            time_s=k[0],
            state=traj.QuadrotorState(
                inertial_from_body=traj.SE3(
                    translation=traj.Vec3(c0=k[1], c1=k[2], c2=k[3]),
                    rotation=traj.SO3(quaternion=traj.Vec4(c0=k[4], c1=k[5], c2=k[6], c3=k[7]))),
                body_velocity=traj.Vec6(c0=k[8], c1=k[9], c2=k[10], c3=k[11], c4=k[12], c5=k[13])),
            control=traj.Vec4(c0=k[14], c1=k[15], c2=k[16], c3=k[17])))
    return traj.QuadrotorTrajectory(points=pts)


def extract_traj_array(trajectory):
    """QuadrotorTrajectory -> (n, 18) array, same columns as the reference's extract_traj_array"""
    return np.asarray(_decode_trajectory(trajectory))


def options_message(o):
    return opts.ILQROptions(
        line_search_params=opts.LineSearchParams(step_update=o["step_update"],
                                                 desired_reduction_frac=o["desired_reduction_frac"],
                                                 max_iters=int(o["ls_max_iters"])),
        convergence_criteria=opts.ConvergenceCriteria(rtol=o["rtol"], atol=o["atol"], max_iters=o["max_iters"]),
        populate_debug=bool(o["populate_debug"]))


def main(horizon_s=4.0, verbose=False):
    cfg = pb.config1(horizon_s)  # demo constants, quadrotor_ilqr.py:257-292
    desired_traj = trajectory_message(cfg["desired"])
    m = cfg["model"]
    ilqr = QuadrotorILQR(m["mass_kg"], m["inertia"], m["arm_length_m"], m["torque_to_thrust_ratio_m"], m["g_mpss"],
                         cfg["Q"], cfg["R"], desired_traj, cfg["dt"], options_message(cfg["options"]))
    opt_traj, debug = ilqr.solve(desired_traj)  # initial = desired, quadrotor_ilqr.py:306
    costs = np.array([d.cost for d in debug.iter_debugs])
    if verbose:
        print(f"{len(desired_traj.points)} knots, {len(costs)} iterations, final cost {costs[-1]:.6f}")
    return dict(desired=cfg["desired"], optimized=extract_traj_array(opt_traj), costs=costs,
                iters=[extract_traj_array(d.trajectory) for d in debug.iter_debugs])


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Quadrotor iLQR demo on the MI355X solver (no plots).")
    ap.add_argument("--horizon_s", type=float, default=4.0)
    main(ap.parse_args().horizon_s, verbose=True)
