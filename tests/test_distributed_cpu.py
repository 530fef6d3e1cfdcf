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


def _worker(rank, world, port, outdir, native=True, reversible=False):
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
    # rank 0 carries the chain's random generator; the other ranks are seeded differently on
    # purpose.  numpy path: parameters are drawn on rank 0 and broadcast; native path:
    # only the chain's base seed comes from rank 0's generator (one broadcast), the draws are a
    # function of (all-reduced statistics, seed, sweep) that every rank evaluates itself
    np.random.seed(5 if rank == 0 else 1000 + rank)
    sampler = bhmm_amd.BayesianHMMSampler(obs, 3, initial_model=hmm, reversible=reversible,
                                          engine_factory=OracleEngine, native_parameters=native,
                                          transition_matrix_sampling_steps=20)
    from bhmm_amd.estimators.bayesian_sampling import _unpack_path_stats
    C, n0, emis = _unpack_path_stats(sampler._updateHiddenStateTrajectories(seed=1, keep_paths=True),
                                     'gaussian', 3, 0)
    first_paths = [np.array(p) for p in sampler.model.hidden_state_trajectories]
    chain = sampler.sample(4, save_hidden_state_trajectory=True)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), L=est.likelihoods,
             chain_A=np.array([m.transition_matrix for m in chain]),
             chain_mu=np.array([m.output_model.means for m in chain]),
             chain_sig=np.array([m.output_model.sigmas for m in chain]),
             chain_pi=np.array([m.initial_distribution for m in chain]),
             chain_p3=np.array(chain[-1].hidden_state_trajectories[3]),
             gp2=first_paths[2],
             A=hmm.transition_matrix, pi=hmm.initial_distribution, mu=hmm.output_model.means,
             sig=hmm.output_model.sigmas, C=est.count_matrix,
             v0=hmm.hidden_state_trajectories[0], v4=hmm.hidden_state_trajectories[4],
             gC=C, gn0=n0, npaths=len([p for p in sampler.model.hidden_state_trajectories
                                       if p is not None]))
    dist.destroy_process_group()


@pytest.mark.parametrize("native,reversible", [(True, False), (True, True), (False, False)])
def test_two_rank_em_equals_single_process(native, reversible):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    import bhmm_amd
    from oracle_engine import OracleEngine
    obs, init = _problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=False,
                                              accuracy=1e-5, maxit=15, engine_factory=OracleEngine)
    ref = est.fit()
    np.random.seed(5)
    sampler = bhmm_amd.BayesianHMMSampler(obs, 3, initial_model=ref, reversible=reversible,
                                          engine_factory=OracleEngine, native_parameters=native,
                                          transition_matrix_sampling_steps=20)
    from bhmm_amd.estimators.bayesian_sampling import _unpack_path_stats
    gC, gn0, _ = _unpack_path_stats(sampler._updateHiddenStateTrajectories(seed=1, keep_paths=True),
                                    'gaussian', 3, 0)
    gp2 = np.array(sampler.model.hidden_state_trajectories[2])
    chain = sampler.sample(4, save_hidden_state_trajectory=True)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, native, reversible), nprocs=2, join=True)
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
        # Gibbs: uniforms are addressed by global position -> the paths of the sharded run are
        # the single-process paths; parameters come from rank 0's generator only -> the whole
        # chain is the single-process chain on every rank (rank 1 was seeded differently)
        assert np.array_equal(r["gC"], gC) and np.array_equal(r["gn0"], gn0)
        assert np.array_equal(r["gp2"], gp2)
        np.testing.assert_allclose(r["chain_A"], [m.transition_matrix for m in chain], rtol=1e-9)
        np.testing.assert_allclose(r["chain_mu"], [m.output_model.means for m in chain], rtol=1e-9)
        np.testing.assert_allclose(r["chain_sig"], [m.output_model.sigmas for m in chain], rtol=1e-9)
        np.testing.assert_allclose(r["chain_pi"], [m.initial_distribution for m in chain],
                                   rtol=1e-9, atol=1e-15)
        assert np.array_equal(r["chain_p3"], chain[-1].hidden_state_trajectories[3])
    assert np.array_equal(r0["gC"], r1["gC"])                # integer all-reduce: identical


def _worker_one_traj(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import bhmm_amd
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    obs, init = _problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs[:1], 3, initial_model=init, reversible=False,
                                              accuracy=1e-5, maxit=5, engine_factory=OracleEngine)
    hmm = est.fit()
    np.savez(os.path.join(outdir, "one%d.npz" % rank), L=est.likelihoods, nloc=len(est.local_trajectories),
             v=hmm.hidden_state_trajectories[0])
    dist.destroy_process_group()


def test_more_ranks_than_trajectories():
    """A rank whose shard is empty contributes zeros and still takes part in every collective."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    import bhmm_amd
    from oracle_engine import OracleEngine
    obs, init = _problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs[:1], 3, initial_model=init, reversible=False,
                                              accuracy=1e-5, maxit=5, engine_factory=OracleEngine)
    ref = est.fit()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_one_traj, args=(2, _free_port(), d), nprocs=2, join=True)
        rs = [np.load(os.path.join(d, "one%d.npz" % r)) for r in range(2)]
    assert sorted(int(r["nloc"]) for r in rs) == [0, 1]
    for r in rs:
        np.testing.assert_allclose(r["L"], est.likelihoods, rtol=1e-12)
        assert np.array_equal(r["v"], ref.hidden_state_trajectories[0])
