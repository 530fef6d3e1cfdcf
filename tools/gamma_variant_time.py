"""Time of the gamma-storing (careful) instantiation of k_estep on configs[1]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = torch.from_numpy(synth_gaussian(model, K, T, seed=2000).reshape(-1)).cuda()
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
for sg in (False, True):
    for _ in range(3):
        eng.estep(*args, store_gamma=sg)
    t0 = time.perf_counter()
    for _ in range(20):
        eng.estep(*args, store_gamma=sg)
    dt = (time.perf_counter() - t0) / 20
    print("store_gamma", sg, "%.3f ms per E-step, sweep kernel %.3f ms" % (dt * 1e3, eng.kernel_ms(2)))
