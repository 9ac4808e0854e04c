"""Generates tests/golden/reference_demo_inputs.npz -- REFERENCE-DERIVED vectors (not self-goldens).

Build container only: imports the reference's own demo driver, /root/reference/src/quadrotor_ilqr.py, as a
module (it is never copied, and it never travels to the GPU box: only the .npz it produces does) and records
what ITS functions compute:

  * desired40 / desired100: the box-climb desired trajectory built point by point with the reference's
    make_traj_pt / make_state (quadrotor_ilqr.py:68-106) as the reference's main() builds it (:256-270:
    np.arange(0, horizon_s, dt_s), vel_mps = 10, zero control) and read back with the reference's
    extract_traj_array (:40-65) in the reference's IDX order (:19-37);
  * desired40_wire / desired100_wire: SerializeToString() of those messages (python-protobuf's encoder over this
    repo's descriptor tables, which tests/test_boundary_cpu.py holds to the reference's .proto files);
  * demo_*: every argument the reference's main() (:256-306) hands to QuadrotorILQR(...) and .solve(...), captured by
    a recording stand-in for the solver class (the reference's C++ solver cannot be built here: Eigen / manif absent):
    mass, inertia, arm length, torque-to-thrust ratio, g, Q, R, dt, the options message's fields and wire bytes, the
    desired trajectory it passes as constructor argument and as initial trajectory.

What the import needs that the image lacks: the plotting dependency `stl` (numpy-stl), used only inside
animate_trajectories -- an empty module of that name satisfies the import; nothing of it is called.  The `src` package the
reference imports (`src.trajectory_pb2`, `src.ilqr_options_pb2`, `src.quadrotor_ilqr_binding`) is this repo's, exactly as a user
who switched would have it.

    python tests/golden/make_reference_fixture.py          (from the repo root, in the build container)
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/quadrotor_ilqr.py"
OUT = os.path.join(ROOT, "tests", "golden", "reference_demo_inputs.npz")


def load_reference_driver():
    if not os.path.exists(REF):
        raise SystemExit("the reference is not present (this script runs in the build container only)")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, ROOT)
    stl = types.ModuleType("stl")
    stl.mesh = types.ModuleType("stl.mesh")       # `from stl import mesh`; only animate_trajectories touches it
    sys.modules.setdefault("stl", stl)
    sys.modules.setdefault("stl.mesh", stl.mesh)
    spec = importlib.util.spec_from_file_location("reference_quadrotor_ilqr", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Recorder:
    """Stands where the solver class stands in the reference's main(): keeps the ten positional constructor arguments and
    the initial trajectory, and returns (initial trajectory, empty debug) so that main() runs to its end."""
    calls = []

    def __init__(self, *args):
        assert len(args) == 10, len(args)   # quadrotor_ilqr_binding.cc:27-37
        self.args = args
        _Recorder.calls.append(self)

    def solve(self, initial):
        import src.ilqr_debug_pb2 as dbg
        self.initial = initial
        return initial, dbg.QuadrotorILQRDebug()


def main():
    ref = load_reference_driver()
    import src.trajectory_pb2 as traj
    out = {}
    # the reference's IDX enumeration itself (:19-37): names in value order
    out["idx_names"] = np.array([m.name for m in sorted(ref.IDX, key=int)])

    for name, horizon_s in (("desired40", 4.0), ("desired100", 10.0)):
        dt_s, vel_mps = 0.1, 10                                  # :257, :260
        msg = traj.QuadrotorTrajectory(points=[
            traj.QuadrotorTrajectoryPoint(time_s=t_s, state=ref.make_traj_pt(t_s, vel_mps, horizon_s), control=traj.Vec4())
            for t_s in np.arange(0, horizon_s, dt_s)])           # :259, :261-270
        out[name] = ref.extract_traj_array(msg)
        out[name + "_wire"] = np.frombuffer(msg.SerializeToString(), dtype=np.uint8)

    # make_state on general Euler angles (:68-80): the quaternion convention (scipy x,y,z,w -> proto w,x,y,z)
    rng = np.random.default_rng(7)
    eul = rng.uniform(-np.pi, np.pi, (16, 3))
    pos = rng.uniform(-5, 5, (16, 3))
    states = [ref.make_state(*p, *e) for p, e in zip(pos, eul)]
    out["make_state_xyz_rpy"] = np.hstack([pos, eul])
    out["make_state_pose"] = np.array([[s.inertial_from_body.translation.c0, s.inertial_from_body.translation.c1,
                                        s.inertial_from_body.translation.c2,
                                        s.inertial_from_body.rotation.quaternion.c0, s.inertial_from_body.rotation.quaternion.c1,
                                        s.inertial_from_body.rotation.quaternion.c2, s.inertial_from_body.rotation.quaternion.c3]
                                       for s in states])

    # the reference's main() with the solver class replaced by the recorder
    ref.QuadrotorILQR = _Recorder
    ref.main(show_plots=False)
    (call,) = _Recorder.calls
    mass, inertia, arm, ratio, g, Q, R, desired, dt, options = call.args
    out.update(demo_mass_kg=np.float64(mass), demo_inertia=np.asarray(inertia, dtype=float), demo_arm_length_m=np.float64(arm),
               demo_torque_to_thrust_ratio_m=np.float64(ratio), demo_g_mpss=np.float64(g), demo_Q=np.asarray(Q, dtype=float),
               demo_R=np.asarray(R, dtype=float), demo_dt_s=np.float64(dt),
               demo_desired=ref.extract_traj_array(desired), demo_initial=ref.extract_traj_array(call.initial),
               demo_options=np.array([options.line_search_params.step_update, options.line_search_params.desired_reduction_frac,
                                      options.line_search_params.max_iters, options.convergence_criteria.rtol,
                                      options.convergence_criteria.atol, options.convergence_criteria.max_iters,
                                      float(options.populate_debug)]),
               demo_options_wire=np.frombuffer(options.SerializeToString(), dtype=np.uint8))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
