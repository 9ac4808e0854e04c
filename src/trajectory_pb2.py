"""src/trajectory.proto of the reference (lines 5-50), built without protoc."""
from src._proto_build import DOUBLE, MESSAGE, build_file


def _vec(n):
    return [("c%d" % i, i + 1, DOUBLE, None, False) for i in range(n)]


DESCRIPTOR, _m = build_file("src/trajectory.proto", [
    ("Vec3", _vec(3)),
    ("Vec4", _vec(4)),
    ("Vec6", _vec(6)),
    ("SO3", [("quaternion", 1, MESSAGE, "Vec4", False)]),  # coefficients w, x, y, z
    ("SE3", [("translation", 1, MESSAGE, "Vec3", False), ("rotation", 2, MESSAGE, "SO3", False)]),
    ("QuadrotorState", [("inertial_from_body", 1, MESSAGE, "SE3", False),
                        ("body_velocity", 2, MESSAGE, "Vec6", False)]),
    ("QuadrotorTrajectoryPoint", [("time_s", 1, DOUBLE, None, False),
                                  ("state", 2, MESSAGE, "QuadrotorState", False),
                                  ("control", 3, MESSAGE, "Vec4", False)]),
    ("QuadrotorTrajectory", [("points", 1, MESSAGE, "QuadrotorTrajectoryPoint", True)]),
])
Vec3 = _m["Vec3"]
Vec4 = _m["Vec4"]
Vec6 = _m["Vec6"]
SO3 = _m["SO3"]
SE3 = _m["SE3"]
QuadrotorState = _m["QuadrotorState"]
QuadrotorTrajectoryPoint = _m["QuadrotorTrajectoryPoint"]
QuadrotorTrajectory = _m["QuadrotorTrajectory"]
