"""End-to-end checks from raw data (no initial model given): heuristic starts, Baum-Welch and Gibbs
sampling on the GPU, against the generating models of synthetic data.  Prints one line per scenario.
    python tools/e2e_scenarios.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bhmm_amd as bhmm  # noqa: E402
from bhmm_amd.util.testsystems import generate_transition_matrix  # noqa: E402


def gaussian(n, K, T, seed, **kw):
    rs = np.random.RandomState(seed)
    model, O, S = bhmm.testsystems.generate_synthetic_observations(nstates=n, ntrajectories=K, length=T,
                                                                   rng=rs, **kw)
    t = time.perf_counter()
    h = bhmm.estimate_hmm(O, n, multi_start=True)
    dt = time.perf_counter() - t
    agree = np.mean(np.concatenate(h.hidden_state_trajectories) == np.concatenate(S))
    print("gaussian n=%2d %4d x %6d: fit %5.1f s  mu err %.3f  sigma err %.3f  T err %.3f  viterbi %.3f" % (
        n, K, T, dt, np.abs(h.output_model.means - model.output_model.means).max(),
        np.abs(h.output_model.sigmas - model.output_model.sigmas).max(),
        np.abs(h.transition_matrix - model.transition_matrix).max(), agree))
    return model, O, S, h


def discrete(n, M, K, T, seed):
    rs = np.random.RandomState(seed)
    P = generate_transition_matrix(n, rng=rs)
    B = np.zeros((n, M))
    w = M // n
    for i in range(n):
        B[i, w * i:w * i + w] = np.arange(w, 0, -1) / float(w * (w + 1) // 2)
    B = 0.95 * B + 0.05 / M
    truth = bhmm.discrete_hmm(np.ones(n) / n, P, B)
    O, S = truth.generate_synthetic_observation_trajectories(K, T, rng=rs)
    t = time.perf_counter()
    h = bhmm.estimate_hmm(O, n)
    dt = time.perf_counter() - t
    Bh = h.output_model.output_probabilities
    perm = np.argsort(Bh.argmax(axis=1))
    agree = np.mean(perm.argsort()[np.concatenate(h.hidden_state_trajectories)] == np.concatenate(S))
    print("discrete n=%2d M=%2d %3d x %6d: fit %5.1f s  T err %.3f  B err %.3f  viterbi %.3f" % (
        n, M, K, T, dt, np.abs(h.transition_matrix[np.ix_(perm, perm)] - P).max(),
        np.abs(Bh[perm] - B).max(), agree))


def main():
    gaussian(2, 10, 30000, 1)
    model, O, S, h = gaussian(3, 8, 20000, 0)
    np.random.seed(2)
    s = bhmm.bayesian_hmm(O, h, nsample=30)
    print("   Gibbs, 30 samples: slowest timescale %.1f +- %.2f (ML %.1f), means within %.3f of ML" % (
        s.timescales_mean[0], s.timescales_std[0], h.timescales[0],
        np.abs(s.means_mean - h.output_model.means).max()))
    gaussian(4, 10, 30000, 2)          # the mixture start alone ends in a poor optimum here
    gaussian(5, 10, 30000, 3)
    gaussian(3, 1, 10000, 9)
    gaussian(3, 400, 50, 11)
    gaussian(10, 16, 50000, 10, omin=-30, omax=30, sigma_min=0.6, sigma_max=1.2)   # 9..64 states
    gaussian(20, 16, 50000, 20, omin=-60, omax=60, sigma_min=0.6, sigma_max=1.2)
    discrete(3, 12, 6, 30000, 7)
    discrete(10, 50, 12, 40000, 5)


if __name__ == "__main__":
    main()
