"""Run a few E-steps plus a calibration copy of known size (for the FETCH_SIZE / WRITE_SIZE
counter calibration prescribed by MI355X_MICROARCH.md, section HBM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = torch.from_numpy(synth_gaussian(model, K, T, seed=2000).reshape(-1)).cuda()
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
for _ in range(3):
    eng.estep(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()   # 1 GiB
for _ in range(3):
    y = x.clone()                                                        # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("done")
