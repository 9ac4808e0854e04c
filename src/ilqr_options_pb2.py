"""src/ilqr_options.proto of the reference (lines 5-21), built without protoc.
ConvergenceCriteria.max_iters is a double and LineSearchParams.max_iters an int32, as there."""
from src._proto_build import BOOL, DOUBLE, INT32, MESSAGE, build_file

DESCRIPTOR, _m = build_file("src/ilqr_options.proto", [
    ("LineSearchParams", [("step_update", 1, DOUBLE, None, False),
                          ("desired_reduction_frac", 2, DOUBLE, None, False),
                          ("max_iters", 3, INT32, None, False)]),
    ("ConvergenceCriteria", [("rtol", 1, DOUBLE, None, False), ("atol", 2, DOUBLE, None, False),
                             ("max_iters", 3, DOUBLE, None, False)]),
    ("ILQROptions", [("line_search_params", 1, MESSAGE, "LineSearchParams", False),
                     ("convergence_criteria", 2, MESSAGE, "ConvergenceCriteria", False),
                     ("populate_debug", 3, BOOL, None, False)]),
])
LineSearchParams = _m["LineSearchParams"]
ConvergenceCriteria = _m["ConvergenceCriteria"]
ILQROptions = _m["ILQROptions"]
