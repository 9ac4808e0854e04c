"""Observed maxima of the solver-against-oracle differences, printed by the GPU tests (run with -s or read the captured output;
`python -m pytest tests -m gpu -q -rP` lists them) so that every bar in tests/ can be read against what was measured: the bars are
the stated tolerances of SURVEY.md section 8(c) (cost 1e-9 relative, trajectory 1e-6) wherever the measurement allows, and ten
times the observed maximum, stated in the test, where it does not.  Appends to gpurun_out/observed.txt when
that directory exists (the GPU box), so the numbers come back with the run."""
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def observed(label, out, ref):
    rel = float(np.max(np.abs(np.asarray(out["cost"]) - np.asarray(ref["cost"])) / np.abs(np.asarray(ref["cost"]))))
    tr = float(np.max(np.abs(np.asarray(out["traj"]) - np.asarray(ref["traj"]))))
    line = f"[observed] {label}: max rel cost difference {rel:.3e}, max abs trajectory difference {tr:.3e}"
    print(line)
    d = os.path.join(_ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "observed.txt"), "a") as f:
            f.write(line + "\n")
    return rel, tr
