"""The diagnostics build of the library (quadrotorilqr_amd/lib/libquadrotor_ilqr_diag.so, `make -C quadrotorilqr_amd/csrc diag`,
-DQILQR_DIAG) as a second, independent instance of the ctypes binding: the kernels that measured behind the product's and were
taken out of it (k_solve4: persistent = 1; k_backward2: force_general = 3) keep their parity tests, and the bounded spins of
k_rollout16 get a fault to trip over.  Test infrastructure: nothing in the product loads this library."""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_mod = None


def capi_diag():
    global _mod
    if _mod is None:
        path = os.path.join(_ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_diag.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C quadrotorilqr_amd/csrc diag` (python -c 'import __graft_entry__ as g; g.build()' does)")
        spec = importlib.util.spec_from_file_location("quadrotorilqr_amd_capi_diag", os.path.join(_ROOT, "quadrotorilqr_amd", "capi.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.LIB_PATH = path
        _mod = m
    return _mod
