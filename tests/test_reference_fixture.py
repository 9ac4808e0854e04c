"""The workload definitions, the demo driver's message builder and the wire codec against REFERENCE-DERIVED vectors:
tests/golden/reference_demo_inputs.npz, written by tests/golden/make_reference_fixture.py from the reference's own Python
(quadrotor_ilqr.py:19-106 make_state / make_traj_pt / extract_traj_array / IDX, and the arguments its main() :256-306 hands to the
solver).  Unlike oracle_golden.npz these are not self-goldens: the numbers were computed by the reference's code."""
import glob
import os

import numpy as np
import pytest

from quadrotorilqr_amd import problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.load(os.path.join(ROOT, "tests", "golden", "reference_demo_inputs.npz"))


@pytest.fixture(scope="module")
def binding():
    if not glob.glob(os.path.join(ROOT, "src", "quadrotor_ilqr_binding*.so")):
        import __graft_entry__ as g
        g.build()
    import src.quadrotor_ilqr_binding as b
    return b


def test_knot_layout_is_the_reference_enumeration():
    """quadrotor_ilqr.py:19-37: the 18 columns every array of the C ABI uses (include/quadrotor_ilqr.h, problems.PT)"""
    assert list(F["idx_names"]) == [
        "time_s", "translation_x_m", "translation_y_m", "translation_z_m", "quaternion_w", "quaternion_x", "quaternion_y",
        "quaternion_z", "vel_translational_x_mps", "vel_translational_y_mps", "vel_translational_z_mps", "vel_rotational_x_radps",
        "vel_rotational_y_radps", "vel_rotational_z_radps", "control_0", "control_1", "control_2", "control_3"]
    assert len(F["idx_names"]) == pb.PT


@pytest.mark.parametrize("name,horizon_s", [("desired40", 4.0), ("desired100", 10.0)])
def test_box_climb_desired_is_the_reference_trajectory_bit_for_bit(name, horizon_s):
    """configs[0]'s input (100 knots) and the demo's (40): problems.box_climb_desired against make_traj_pt / make_state"""
    ours = pb.box_climb_desired(horizon_s)
    assert ours.shape == F[name].shape
    np.testing.assert_array_equal(ours, F[name])


def test_demo_constants_are_what_the_reference_main_passes():
    """quadrotor_ilqr.py:256-306: every argument of QuadrotorILQR(...) and of .solve(...)"""
    m = pb.MODEL_D
    assert m["mass_kg"] == F["demo_mass_kg"] and m["arm_length_m"] == F["demo_arm_length_m"]
    assert m["torque_to_thrust_ratio_m"] == F["demo_torque_to_thrust_ratio_m"] and m["g_mpss"] == F["demo_g_mpss"]
    np.testing.assert_array_equal(m["inertia"], F["demo_inertia"])
    np.testing.assert_array_equal(pb.Q_DEMO, F["demo_Q"])
    np.testing.assert_array_equal(pb.R_DEMO, F["demo_R"])
    assert pb.DT_DEMO == F["demo_dt_s"]
    o = pb.OPTIONS_DEMO
    assert [o["step_update"], o["desired_reduction_frac"], o["ls_max_iters"], o["rtol"], o["atol"], o["max_iters"],
            float(o["populate_debug"])] == list(F["demo_options"])
    np.testing.assert_array_equal(F["demo_desired"], F["desired40"])      # main() uses horizon_s = 4.0
    np.testing.assert_array_equal(F["demo_initial"], F["demo_desired"])    # and starts from the desired trajectory (:306)
    cfg = pb.config1(4.0)
    np.testing.assert_array_equal(cfg["desired"], F["demo_desired"])
    np.testing.assert_array_equal(cfg["init"][0], F["demo_initial"])


def test_make_state_quaternion_convention():
    """quadrotor_ilqr.py:68-80: Euler 'xyz' -> scipy (x, y, z, w) -> proto (w, x, y, z).  problems.se3_exp on a pure roll is the
    same rotation; a general attitude is checked through the rotation it represents (q and -q are the same attitude)."""
    from scipy.spatial.transform import Rotation
    xyzrpy, pose = F["make_state_xyz_rpy"], F["make_state_pose"]
    np.testing.assert_array_equal(pose[:, :3], xyzrpy[:, :3])
    q = pose[:, 3:]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-15)
    Rm = Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()
    want = Rotation.from_euler("xyz", xyzrpy[:, 3:]).as_matrix()
    np.testing.assert_allclose(Rm, want, atol=1e-14)
    roll = 2.0 * np.pi / 3.0                       # the demo's third leg
    ours = pb.se3_exp([0, 0, 0, roll, 0, 0])[3:]
    ref = F["desired40"][20, 4:8]
    np.testing.assert_allclose(ours, ref, atol=1e-16)


@pytest.mark.parametrize("name", ["desired40", "desired100"])
def test_message_builder_and_codec_against_reference_built_messages(name, binding):
    """src.demo.trajectory_message(array) serialises to the very bytes of the message the reference's make_traj_pt built, and
    the C++ wire codec (src/proto_wire.h) decodes those bytes to the reference's extract_traj_array output and encodes them
    back byte for byte."""
    import src.trajectory_pb2 as traj
    from src.demo import extract_traj_array, trajectory_message
    wire = F[name + "_wire"].tobytes()
    arr = F[name]
    assert trajectory_message(arr).SerializeToString() == wire
    msg = traj.QuadrotorTrajectory.FromString(wire)
    np.testing.assert_array_equal(extract_traj_array(msg), arr)
    np.testing.assert_array_equal(np.asarray(binding._decode_trajectory(msg)), arr)
    assert binding._encode_trajectory(arr).SerializeToString() == wire


def test_options_message_is_the_reference_main_options(binding):
    from src.demo import options_message
    import src.ilqr_options_pb2 as opts
    wire = F["demo_options_wire"].tobytes()
    assert options_message(pb.OPTIONS_DEMO).SerializeToString() == wire
    o = opts.ILQROptions.FromString(wire)
    assert o.populate_debug and o.line_search_params.max_iters == 100 and o.convergence_criteria.max_iters == 100


def test_fixture_generator_does_not_travel():
    """the generator reads /root/reference; nothing else under tests/ or the product does at run time"""
    src = open(os.path.join(ROOT, "tests", "golden", "make_reference_fixture.py")).read()
    assert "/root/reference/src/quadrotor_ilqr.py" in src
    for path in glob.glob(os.path.join(ROOT, "tests", "test_*.py")) + glob.glob(os.path.join(ROOT, "quadrotorilqr_amd", "*.py")):
        if os.path.basename(path) == "test_reference_fixture.py":
            continue
        text = open(path).read()
        assert "spec_from_file_location(\"reference" not in text
