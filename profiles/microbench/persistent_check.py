import sys, time, numpy as np
sys.path.insert(0, '.')
from quadrotorilqr_amd import capi, problems as pb
for B, N in [(4, 20), (7, 33), (64, 60), (1024, 100)]:
    cfg = pb.config2(B=B, N=N)
    a = capi.from_config(cfg, persistent=2).solve_batch(cfg["init"])
    t0 = time.time()
    b = capi.from_config(cfg, persistent=1).solve_batch(cfg["init"])
    print(B, N, "persistent solve returned in %.3f s" % (time.time() - t0), flush=True)
    for k in ("status", "iters", "n_bwd", "n_fwd"):
        print("  ", k, "equal:", np.array_equal(a[k], b[k]), end="")
    print()
    print("   cost max rel diff %.3e  traj max abs diff %.3e" % (np.max(np.abs(a["cost"] - b["cost"]) / np.abs(a["cost"])), np.abs(a["traj"] - b["traj"]).max()), flush=True)
