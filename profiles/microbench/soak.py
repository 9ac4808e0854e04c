import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
bad = 0
for seed in range(12, 132):
    try:
        T.test_randomised_models_and_horizons_match_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e).splitlines()[:6])
print("soak done, failures:", bad)
