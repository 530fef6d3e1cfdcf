"""E-step time against the chunk length for batches smaller than configs[1] (8-state Gaussian): is the
automatic plan (32768 chunks whatever the warm-up) the best one when chunks get shorter than the warm-up?
python tools/chunk_scan.py K T"""
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
base = None
for mult in (0, 1, 1.5, 2, 3, 4, 6):
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8,
                                chunk=0 if mult == 0 else int(round(base * mult)))
    def em_like():
        eng.estep_launch(*margs)
        eng.estep_fetch_packed()
    for _ in range(5):
        em_like()
    dt = timeit(em_like, 5, eng.sync)
    if base is None:
        base = eng.chunk_len
    print("K=%d T=%d chunk %s: %d chunks x %d, W %d, E-step %.3f ms (sweep %.3f)" % (
        K, T, "auto" if mult == 0 else "x%g" % mult, eng.num_chunks, eng.chunk_len, eng.get_option("spec_W"),
        dt * 1e3, eng.kernel_ms(2)))
    eng.close()
