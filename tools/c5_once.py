"""configs[4] (Gibbs hidden-path sweep, 8-state Gaussian 256 x 1e5) a few times: for kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian_device
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import timeit
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = synth_gaussian_device(model, K, T, seed=11, device="cuda:0")
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
for _ in range(3):
    eng.sample_paths(*args, seed=1, want_paths=False)
dt = timeit(lambda: eng.sample_paths(*args, seed=1, want_paths=False), 5)
print("ms %.3f" % (dt * 1e3))
pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
dt = timeit(lambda: eng.viterbi_u8(*args, out=pdev), 5)
print("viterbi (uint8 paths, device-resident) ms %.3f" % (dt * 1e3))
ppin = torch.empty(K * T, dtype=torch.uint8).pin_memory()
dt = timeit(lambda: eng.viterbi_u8(*args, out=ppin), 5)
print("viterbi (uint8 paths, pinned host) ms %.3f" % (dt * 1e3))
