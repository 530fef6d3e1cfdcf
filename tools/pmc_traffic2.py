"""Workload for the HBM-traffic PMC passes (tools/profile_round2.sh): E-steps, Gibbs hidden-path
sweeps and Viterbi on BASELINE configs[1]'s shape, plus a calibration copy of known size (the
FETCH_SIZE / WRITE_SIZE calibration MI355X_MICROARCH.md, section HBM, prescribes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_c2_model, synth_gaussian
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = torch.from_numpy(synth_gaussian(model, K, T, seed=2000).reshape(-1)).cuda()
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
margs = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
for _ in range(3):
    eng.estep(*margs)
sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda")
for _ in range(3):
    eng.sample_paths_dev(*margs, sbuf.data_ptr(), seed=1)
pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda")
for _ in range(3):
    eng.viterbi_u8(*margs, out=pdev)
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()   # 1 GiB
for _ in range(3):
    y = x.clone()                                                        # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("done")
