"""Wall time of whole EM iterations (E-step on the GPU + host M-step) of MaximumLikelihoodEstimator
on the configs[1] shape, split into its parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bhmm_amd
from bench import make_c2_model, synth_gaussian
K, T = 256, 100000
model = make_c2_model()
obs2 = synth_gaussian(model, K, T, seed=2000)
obs = [obs2[k] for k in range(K)]
init = bhmm_amd.gaussian_hmm(model["pi"], model["A_eval"], model["mu_eval"], model["sigma"])
for rev in (False, True):
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 8, initial_model=init, reversible=rev, accuracy=-1.0, maxit=30)
    t0 = time.perf_counter()
    est.fit()
    dt = time.perf_counter() - t0
    print("reversible", rev, "iterations", est.count_it if hasattr(est, "count_it") else len(est.likelihoods),
          "total %.1f ms -> %.2f ms per EM iteration" % (dt * 1e3, dt * 1e3 / max(1, len(est.likelihoods))))
