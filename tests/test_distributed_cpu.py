"""N > 1 path on CPU: two processes over gloo.  Trajectories are LPT-partitioned, each rank
runs its shard's E-step (oracle-backed test double standing in for the GPU engine), the packed
statistics are all-reduced, and every rank performs the same M-step -- results must equal the
single-process run (up to summation order)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    sys.path.insert(0, HERE)
    from test_host_logic import _gauss_problem
    return _gauss_problem(seed=3, K=5, T=250)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import bhmm_amd
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    obs, init = _problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=False,
                                              accuracy=1e-5, maxit=15, engine_factory=OracleEngine)
    assert est._comm.world == world and sorted(sum(est._parts, [])) == list(range(len(obs)))
    hmm = est.fit()
    np.random.seed(5)
    sampler = bhmm_amd.BayesianHMMSampler(obs, 3, initial_model=hmm, reversible=False,
                                          engine_factory=OracleEngine)
    C, n0, emis = sampler._updateHiddenStateTrajectories(seed=1, keep_paths=True)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), L=est.likelihoods,
             A=hmm.transition_matrix, pi=hmm.initial_distribution, mu=hmm.output_model.means,
             sig=hmm.output_model.sigmas, C=est.count_matrix,
             v0=hmm.hidden_state_trajectories[0], v4=hmm.hidden_state_trajectories[4],
             gC=C, gn0=n0, npaths=len([p for p in sampler.model.hidden_state_trajectories
                                       if p is not None]))
    dist.destroy_process_group()


def test_two_rank_em_equals_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    import bhmm_amd
    from oracle_engine import OracleEngine
    obs, init = _problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=False,
                                              accuracy=1e-5, maxit=15, engine_factory=OracleEngine)
    ref = est.fit()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0 = np.load(os.path.join(d, "rank0.npz"))
        r1 = np.load(os.path.join(d, "rank1.npz"))
    for r in (r0, r1):                                       # every rank holds the full result
        assert len(r["L"]) == len(est.likelihoods)
        np.testing.assert_allclose(r["L"], est.likelihoods, rtol=1e-12)
        np.testing.assert_allclose(r["A"], ref.transition_matrix, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["pi"], ref.initial_distribution, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["mu"], ref.output_model.means, rtol=1e-10)
        np.testing.assert_allclose(r["sig"], ref.output_model.sigmas, rtol=1e-9)
        np.testing.assert_allclose(r["C"], est.count_matrix, rtol=1e-10)
        assert np.array_equal(r["v0"], ref.hidden_state_trajectories[0])
        assert np.array_equal(r["v4"], ref.hidden_state_trajectories[4])
        assert int(r["npaths"]) == 5
        assert r["gC"].sum() == sum(len(o) - 1 for o in obs) and r["gn0"].sum() == 5
    assert np.array_equal(r0["gC"], r1["gC"])                # integer all-reduce: identical
