"""configs[3] shape (64-state Gaussian, 128 x 1e5): where a whole EM iteration of MaximumLikelihoodEstimator
spends its time -- E-step (kernel events / wall), M-step (native), the rest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bhmm_amd
from bench import metastable_matrix, stationary
from bhmm_amd.engine import synth_observations

dev = torch.device("cuda", 0)
rng = np.random.default_rng(64)
n, K, T = int(os.environ.get("C4_N", 64)), 128, int(os.environ.get("C4_T", 100000))
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sig, K, T, seed=6400, device=0)
host = obs.cpu().numpy().reshape(K, T)
init = bhmm_amd.gaussian_hmm(pi, 0.9 * A + 0.1 / n, mu + 0.05, sig)
est = bhmm_amd.MaximumLikelihoodEstimator([host[k] for k in range(K)], n, initial_model=init, reversible=False, device=0)
eng = est._engine
for _ in range(4):
    est.em_step()
te = tm = 0.0
kms = np.zeros(5)
iters = int(os.environ.get("ITERS", 10))
Ws, devs, lls = [], [], []
torch.cuda.synchronize()
t00 = time.perf_counter()
for _ in range(iters):
    t0 = time.perf_counter()
    res = est._estep()
    t1 = time.perf_counter()
    est._update_model(res, maxiter=est._maxit_P)
    t2 = time.perf_counter()
    te += t1 - t0; tm += t2 - t1
    kms += eng.kernel_ms_all()
    Ws.append((int(eng.get_option("spec_W")), int(eng.get_option("carry_W")))); devs.append(eng.get_option("spec_last_dev")); lls.append(res.loglik)
tot = time.perf_counter() - t00
print("n=%d: EM iteration %.3f ms = E-step %.3f (kernel events: fwd %.3f, bwd+stats %.3f, total %.3f) + M-step %.3f ms"
      % (n, 1e3 * tot / iters, 1e3 * te / iters, kms[0] / iters, kms[2] / iters, kms[4] / iters, 1e3 * tm / iters))
print("   (W, carried W) per iteration:", Ws)
print("   boundary deviation:", ["%.1e" % d for d in devs], "spec ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"),
      "tile", eng.get_option("tile"), "trouble", eng.get_option("wide_trouble"))
print("   loglik:", ["%.6f" % l for l in lls[:3]], "...", "%.6f" % lls[-1])
