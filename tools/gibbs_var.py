import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian_device, timeit
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = synth_gaussian_device(model, K, T, seed=2000, device="cuda:0")
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
for _ in range(3): eng.estep(*args)
sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
for rep in range(6):
    if rep % 2 == 1:
        eng.viterbi_u8(*args, out=pdev)
    f0 = eng.get_option("spec_fail"), eng.get_option("spec_ok")
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); eng.sample_paths_dev(*args, sbuf.data_ptr(), seed=1); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(rep, ["%.2f" % t for t in ts], f0, (eng.get_option("spec_fail"), eng.get_option("spec_ok")), eng.get_option("spec_W"))
