"""configs[1] data of bench.py: chunk-parallel Viterbi (8 states) -- the warm-up the search settles at, and the time
against a forced warm-up and chunk length.   python tools/c1_vit_scan.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import make_c2_model, timeit, NSTATES
from bhmm_amd.engine import Engine, synth_observations
dev = torch.device("cuda", 0)
model = make_c2_model()
n, K, T = NSTATES, 256, 100000
buf = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", buf.data_ptr(), model["A"], model["pi"], model["mu"], model["sigma"], K, T, seed=2000, device=0)
margs = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
out = torch.empty(K * T, dtype=torch.uint8, device=dev)
for chunk in (0, 1564, 3128):
    eng = Engine(0)
    eng.set_observations_device("gaussian", buf.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, chunk=chunk)
    eng.estep(*margs)
    for _ in range(8):
        eng.viterbi_u8(*margs, out=out)
    dt = timeit(lambda: eng.viterbi_u8(*margs, out=out), 5, eng.sync, batches=5)
    print("chunk %5d (%d chunks of %d): %.3f ms, W settled at %d (E-step W %d), chunked %d"
          % (chunk, eng.num_chunks, eng.chunk_len, 1e3 * dt, eng.get_option("viterbi_W"), eng.get_option("spec_W"), eng.get_option("viterbi_chunked")), flush=True)
    for W in (64, 96, 128, 192, 280):
        def run():
            eng.set_option("viterbi_W", W)
            eng.viterbi_u8(*margs, out=out)
        run()
        dt = timeit(run, 5, eng.sync, batches=3)
        print("      forced W %4d: %.3f ms chunked %d close %d" % (W, 1e3 * dt, eng.get_option("viterbi_chunked"), eng.get_option("viterbi_close")), flush=True)
    eng.close()
