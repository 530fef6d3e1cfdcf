"""Many short trajectories (the other extreme of the BASELINE shapes): K x T = 65536 x 128 and
8192 x 1000, 8-state Gaussian: E-step / Gibbs sweep / Viterbi time, parity of a few trajectories."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_c2_model, timeit
from bhmm_amd.engine import Engine, synth_observations
from oracle import oracle as orc
m = make_c2_model()
margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
SHAPES = ((65536, 128), (8192, 1000), (1000000, 20))
if os.environ.get("SHAPE"):
    SHAPES = (tuple(int(x) for x in os.environ["SHAPE"].split("x")),)
for K, T in SHAPES:
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=K)
    eng = Engine(0)
    t0 = time.perf_counter()
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    t_set = time.perf_counter() - t0
    r = eng.estep(*margs)
    def em_like():          # what an EM iteration fetches: the packed statistics only
        eng.estep_launch(*margs)
        eng.estep_fetch_packed()
    dt = timeit(em_like, 5, eng.sync)
    o = obs[: 3 * T].cpu().numpy().reshape(3, T)
    ref = orc.estep("gaussian", list(o), *margs)
    err = np.max(np.abs((r.logL_k[:3] - ref["logL"]) / ref["logL"]))
    sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
    dg = timeit(lambda: eng.sample_paths_dev(*margs, sbuf.data_ptr(), seed=1), 5, eng.sync)
    pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
    dv = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 3, eng.sync)
    p0 = pdev[:T].cpu().numpy()
    vok = np.array_equal(p0, orc.viterbi(m["A_eval"], orc.pobs_gaussian(o[0], m["mu_eval"], m["sigma"]), m["pi"]))
    print("K=%d T=%d: set_observations %.3f s, chunks %d x %d, E-step %.3f ms (%.2e steps/s), Gibbs %.3f ms, Viterbi %.3f ms (chunked %d), logL rel err %.1e, viterbi ok %s, spec %s"
          % (K, T, t_set, eng.num_chunks, eng.chunk_len, dt * 1e3, K * T / dt, dg * 1e3, dv * 1e3,
             eng.get_option("viterbi_chunked"), err, vok,
             {k: eng.get_option(k) for k in ("spec_W", "spec_ok", "spec_fail")}))
    eng.close()
    del obs, pdev
    torch.cuda.empty_cache()
