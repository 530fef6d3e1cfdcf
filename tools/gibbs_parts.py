"""Where a Gibbs sweep of the configs[4] chain goes at the full batch (256 x 1e5) and at the shard one of eight
ranks holds (32 x 1e5): the path step alone, the whole sweep, and the host part in between.
   python tools/gibbs_parts.py [K ...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bhmm_amd
from bench import make_c2_model, timeit, NSTATES
from bhmm_amd.engine import Engine, synth_observations
from bhmm_amd.estimators import _tmatrix
dev = torch.device("cuda", 0)
model = make_c2_model()
n, T = NSTATES, 100000
for K in [int(a) for a in sys.argv[1:]] or [256, 32]:
    buf = torch.empty(K * T, dtype=torch.float64, device=dev)
    synth_observations("gaussian", buf.data_ptr(), model["A"], model["pi"], model["mu"], model["sigma"], K, T, seed=2000, device=0)
    eng = Engine(0)
    eng.set_observations_device("gaussian", buf.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
    sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device=dev)
    dt = timeit(lambda: eng.sample_paths_dev(*margs, sbuf.data_ptr(), seed=1), 10, eng.sync, batches=5)
    dt2 = timeit(lambda: eng.sample_paths(*margs, seed=1, want_paths=False), 10, eng.sync, batches=5)
    print("K=%d: path step (stats on device) %.3f ms | (stats to host) %.3f ms | chunks %d x %d steps, W %d"
          % (K, 1e3 * dt, 1e3 * dt2, eng.num_chunks, eng.chunk_len, eng.get_option("spec_W")), flush=True)
    eng.close()
    host = buf.cpu().numpy().reshape(K, T)
    obs = [host[k] for k in range(K)]
    pi, A_eval = model["pi"], model["A_eval"]
    A_rev = _tmatrix.mle_reversible(pi[:, None] * A_eval, maxerr=1e-14)
    for rev, nsteps in ((False, 1000), (True, 1000)):
        init = bhmm_amd.gaussian_hmm(pi, A_rev if rev else A_eval, model["mu_eval"], model["sigma"])
        smp = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=init, reversible=rev,
                                          transition_matrix_sampling_steps=nsteps, device=0)
        smp.sample(5, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); smp.sample(40); torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / 40
        # parts
        tp = tq = tc = 0.0
        import copy
        for _ in range(40):
            t0 = time.perf_counter(); packed = smp._updateHiddenStateTrajectories(); t1 = time.perf_counter()
            smp._update_parameters_native(packed, None); t2 = time.perf_counter()
            mc = copy.deepcopy(smp.model); t3 = time.perf_counter()
            tp += t1 - t0; tq += t2 - t1; tc += t3 - t2
        print("   %s: whole sweep %.3f ms = path step + fetch %.3f | parameter draws + model update %.3f | model copy %.3f"
              % ("reversible(1000)" if rev else "non-reversible", 1e3 * whole, 1e3 * tp / 40, 1e3 * tq / 40, 1e3 * tc / 40), flush=True)
        smp._engine.close()
    del buf
