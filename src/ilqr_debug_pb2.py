"""src/ilqr_debug.proto of the reference (lines 7-14), built without protoc."""
import src.trajectory_pb2  # noqa: F401  (dependency: src/trajectory.proto)
from src._proto_build import DOUBLE, MESSAGE, build_file

DESCRIPTOR, _m = build_file("src/ilqr_debug.proto", [
    ("QuadrotorILQRIterDebug", [("trajectory", 1, MESSAGE, "QuadrotorTrajectory", False),
                                ("cost", 2, DOUBLE, None, False)]),
    ("QuadrotorILQRDebug", [("iter_debugs", 1, MESSAGE, "QuadrotorILQRIterDebug", True)]),
], deps=("src/trajectory.proto",))
QuadrotorILQRIterDebug = _m["QuadrotorILQRIterDebug"]
QuadrotorILQRDebug = _m["QuadrotorILQRDebug"]
