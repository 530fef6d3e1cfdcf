"""A few E-steps on configs[1] for rocprofv3 --pmc passes (tools/pmc_sq.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = torch.from_numpy(synth_gaussian(model, K, T, seed=2000).reshape(-1)).cuda()
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8,
                            chunk=int(os.environ.get("CHUNK", "0")))
for _ in range(3):
    eng.estep(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
torch.cuda.synchronize()
print("done")
