"""Per-step wall time and sweep time of the first E-steps on fresh observations (the bench workload):
how long the slow start lasts, and whether it is the kernels or the host."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian_device
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = synth_gaussian_device(model, K, T, seed=2000, device="cuda:0")
stream = torch.cuda.Stream(device="cuda:0")
torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
time.sleep(idle)
rows = []
for i in range(80):
    t0 = time.perf_counter()
    eng.estep_launch(*args)
    eng.estep_fetch()
    eng.sync()
    rows.append(((time.perf_counter() - t0) * 1e3, eng.kernel_ms(2), eng.kernel_ms(3)))
for i in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 25, 30, 40, 50, 60, 79):
    print("step %2d: wall %.3f ms  sweep %.3f  tail %.3f" % (i, *rows[i]))
