"""Is the slow start of a fresh context (tools/step_ramp.py) the GPU's power management or the
context?  Two contexts on the same data: 60 E-steps on A (until steady), then immediately the first
20 steps of B (fresh), then A again."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian_device
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = synth_gaussian_device(model, K, T, seed=2000, device="cuda:0")
off = np.arange(K + 1, dtype=np.int64) * T
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
stream = torch.cuda.Stream(device="cuda:0"); torch.cuda.set_stream(stream)
A = Engine(0, stream=stream.cuda_stream); A.set_observations_device("gaussian", obs.data_ptr(), off, 8)
B = Engine(0, stream=stream.cuda_stream); B.set_observations_device("gaussian", obs.data_ptr(), off, 8)
def run(eng, n):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.estep_launch(*args); eng.estep_fetch(); eng.sync()
        out.append("%.3f" % ((time.perf_counter() - t0) * 1e3))
    return out
print("A first 10:", run(A, 10)); a = run(A, 60); print("A steps 60-70:", a[-10:])
print("B first 20 (fresh context, GPU already loaded):", run(B, 20))
print("A again:", run(A, 5))
time.sleep(2.0)
print("A after 2 s idle:", run(A, 12))
