import numpy as np, sys
sys.path.insert(0,'.')
from quadrotorilqr_amd import capi, problems as pb
cfg = pb.config2(B=8, N=12, seed=3)
six = capi.from_config(cfg, force_general=4)
fused = capi.from_config(cfg, force_general=5)
trajs = six.forward_sim(cfg["init"], np.zeros((8, 12, 52)), 1.0)
g6, t6 = six.backwards_pass(trajs)
g5, t5 = fused.backwards_pass(trajs)
d = np.abs(g6-g5)
print("terms diff", np.abs(t6-t5).max())
for i in range(11,-1,-1):
    print(i, "k diff %.2e  K diff %.2e   |K| %.2e" % (d[:,i,:4].max(), d[:,i,4:].max(), np.abs(g5[:,i,4:]).max()), " K cols differing:", sorted(set(np.nonzero(d[:,i,4:].reshape(8,12,4))[1].tolist())))
