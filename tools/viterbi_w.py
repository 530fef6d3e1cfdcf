"""How long a warm-up does the chunked Viterbi need?  configs[1] shape; for each W: accepted?, close
decisions, largest boundary deviation, time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_c2_model, timeit
from bhmm_amd.engine import Engine, synth_observations
K, T = 256, 100000
m = make_c2_model()
dev = torch.device("cuda", 0)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=2000)
margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
pdev = torch.empty(K * T, dtype=torch.uint8, device=dev)
ref = None
for W in (280, 192, 128, 96, 64, 48, 32, 24, 16, 8):
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    eng.set_option("spec_W", W)
    eng.viterbi_u8(*margs, out=pdev)
    ch, close, dev_ = eng.get_option("viterbi_chunked"), eng.get_option("viterbi_close"), eng.get_option("spec_last_dev")
    dt = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 3, eng.sync, batches=5)
    p = pdev.cpu().numpy().copy()
    if ref is None:
        ref = p
    print("W %4d: chunked %d close %d dev %.2e  %.3f ms  same paths %s" % (W, ch, close, dev_, 1e3 * dt, np.array_equal(p, ref)))
    eng.close()
