"""The shard one of eight ranks holds of the configs[4] / configs[1] batch (32 x 1e5, 8-state Gaussian): Gibbs path step and
E-step against the chunk length (0 = the automatic plan).   python tools/shard_chunk_scan.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import make_c2_model, timeit, NSTATES
from bhmm_amd.engine import Engine, synth_observations
dev = torch.device("cuda", 0)
model = make_c2_model()
n, T, K = NSTATES, 100000, 32
buf = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", buf.data_ptr(), model["A"], model["pi"], model["mu"], model["sigma"], K, T, seed=2000, device=0)
margs = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
for chunk in (0, 98, 130, 196, 260, 392):
    eng = Engine(0)
    eng.set_observations_device("gaussian", buf.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, chunk=chunk)
    sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device=dev)
    dt = timeit(lambda: eng.sample_paths_dev(*margs, sbuf.data_ptr(), seed=1), 10, eng.sync, batches=5)
    de = timeit(lambda: eng.estep(*margs), 10, eng.sync, batches=5)
    print("chunk %4d: path step %.3f ms, E-step %.3f ms | chunks %d x %d, W %d" % (chunk, 1e3 * dt, 1e3 * de, eng.num_chunks, eng.chunk_len, eng.get_option("spec_W")), flush=True)
    eng.close()
