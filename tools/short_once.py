"""One shape of tests/sweeps/many_short.py (K trajectories of T steps, 8-state Gaussian) for kernel
traces: python tools/short_once.py K T"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, timeit
from bhmm_amd.engine import Engine, synth_observations
K, T = int(sys.argv[1]), int(sys.argv[2])
m = make_c2_model()
margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=K)
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
def em_like():
    eng.estep_launch(*margs)
    eng.estep_fetch_packed()
for _ in range(5):
    em_like()
dt = timeit(em_like, 5, eng.sync)
print("K=%d T=%d E-step %.3f ms, kernel_ms %s" % (K, T, dt * 1e3, [round(eng.kernel_ms(i), 4) for i in range(5)]))
