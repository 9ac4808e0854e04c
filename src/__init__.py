"""Drop-in replacement of the reference's Python package `src` for the iLQR hot path:

    from src.quadrotor_ilqr_binding import QuadrotorILQR      # reference: src/quadrotor_ilqr_binding.cc
    import src.ilqr_options_pb2 as opts                        # reference: src/ilqr_options.proto
    import src.trajectory_pb2 as traj                          # reference: src/trajectory.proto
    import src.ilqr_debug_pb2                                  # reference: src/ilqr_debug.proto

exactly as reference src/quadrotor_ilqr.py:14-16 imports them.  The solver behind it is the HIP
library (include/quadrotor_ilqr.h); the message classes are built at import time from descriptors
constructed in Python (no protoc in this environment), with the reference's package name, file
names, message names and field numbers.
"""
