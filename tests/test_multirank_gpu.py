"""N > 1 on the GPU: two fresh processes (started before this pytest process initialised the
GPU, through the launcher of conftest.py), both driving the HIP engine on GPU 0, sharded
trajectories, the E-step's device statistics buffer summed over ranks, Gibbs parameters drawn on
rank 0 and broadcast.  Their results must be the single-process results:
maximum_likelihood.py:271-282 (sums over trajectories) is what the all-reduce replaces."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "multirank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_equal_one_process(launcher):
    with tempfile.TemporaryDirectory() as d:
        one = launcher.run([[sys.executable, WORKER, "0", "1", "0", d]], timeout=600)
        assert one[0]["rc"] == 0, one[0]["out"]
        port = str(_free_port())
        two = launcher.run([[sys.executable, WORKER, str(r), "2", port, d] for r in range(2)],
                           timeout=600)
        for r in two:
            assert r["rc"] == 0, r["out"]
        ref = np.load(os.path.join(d, "w1_r0.npz"))
        ranks = [np.load(os.path.join(d, "w2_r%d.npz" % r)) for r in range(2)]
    assert int(ref["nlocal"]) == 7
    assert sorted(int(r["nlocal"]) for r in ranks) == [3, 4]     # really sharded
    for r in ranks:
        # EM: same likelihood history (summation order differs -> 1e-12), same model, same paths
        assert len(r["L"]) == len(ref["L"])
        np.testing.assert_allclose(r["L"], ref["L"], rtol=1e-11)
        np.testing.assert_allclose(r["A"], ref["A"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(r["mu"], ref["mu"], rtol=1e-9)
        np.testing.assert_allclose(r["sig"], ref["sig"], rtol=1e-9)
        np.testing.assert_allclose(r["C"], ref["C"], rtol=1e-9)
        assert np.array_equal(r["v"], ref["v"])
        # Gibbs: uniforms addressed by global position + parameters from rank 0 only -> the chain
        # of the sharded run is the single-process chain, on every rank
        np.testing.assert_allclose(r["chain_A"], ref["chain_A"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(r["chain_mu"], ref["chain_mu"], rtol=1e-8)
        np.testing.assert_allclose(r["chain_sig"], ref["chain_sig"], rtol=1e-8)
        assert np.array_equal(r["chain_paths"], ref["chain_paths"])
    assert np.array_equal(ranks[0]["chain_paths"], ranks[1]["chain_paths"])


@pytest.mark.gpu
def test_rccl_collectives_one_rank(launcher):
    """The RCCL code path of the estimators (device statistics buffer -> dist.all_reduce on the
    engine's GPU -> one copy; rank-0 parameter broadcast) with the one rank a one-GPU box allows:
    backend "nccl", world size 1, collectives forced on.  Same results as without a process group."""
    with tempfile.TemporaryDirectory() as d:
        r = launcher.run([[sys.executable, WORKER, "0", "1", "0", d],
                          [sys.executable, WORKER, "0", "1", str(_free_port()), d, "nccl"]], timeout=600)
        for x in r:
            assert x["rc"] == 0, x["out"]
        ref = np.load(os.path.join(d, "w1_r0.npz"))
        got = np.load(os.path.join(d, "rccl1_r0.npz"))
    np.testing.assert_array_equal(got["L"], ref["L"])
    np.testing.assert_array_equal(got["A"], ref["A"])
    assert np.array_equal(got["v"], ref["v"])
    np.testing.assert_array_equal(got["chain_A"], ref["chain_A"])
    np.testing.assert_array_equal(got["chain_mu"], ref["chain_mu"])
    assert np.array_equal(got["chain_paths"], ref["chain_paths"])


def _bench_line(launcher, gpus, small, timeout=900):
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + small
    if gpus > 1:
        cmd.append("--oversubscribe")
    r = launcher.run([cmd], timeout=timeout, env=env)[0]
    assert r["rc"] == 0, r["out"]
    line = [ln for ln in r["out"].splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, r["out"]
    return json.loads(line[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(launcher):
    """`python bench.py --gpus 2` without a launcher environment starts two ranks itself and
    reports n_gpus = 2 (here both ranks share the box's one GPU: --oversubscribe, gloo).  `value` is
    configs[2] STRONG-scaled: the trajectories are drawn by global index, so the 2-rank run sees the
    data of the 1-rank run -- same log-likelihood --, the line carries the same roofline /
    cpu_baseline keys at both N, and `secondary` the whole iterations of the estimator classes."""
    small = ["--steps", "3", "--warmup", "1", "--ntraj", "8", "--length", "50000", "--cpu-traj", "2",
             "--c1-ntraj", "16", "--c1-length", "20000", "--c1-steps", "3", "--c1-warmup", "1",
             "--c1-cpu-traj", "2", "--chain-sweeps", "3", "--em-iterations", "3"]
    one, two = [_bench_line(launcher, g, small) for g in (1, 2)]
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["scaling"] == "strong" == one["scaling"]
    for o in (one, two):
        assert abs(o["value"] - 8 * 50000 * 3 / (o["ms_per_step"] * 3e-3)) < 1e-6 * o["value"]
        assert o["config"]["workload"].startswith("configs[2]")
        assert o["roofline"]["alg_bytes_per_timestep"] == 136
        assert o["roofline"]["alg_bytes_per_launch"] == 136 * (8 // o["n_gpus"]) * 50000
        assert 0 < o["roofline"]["frac"] < 1 and o["roofline"]["sweep_kernel_ms"] > 0
        assert o["cpu_baseline"]["kind"] == "reference" and o["cpu_baseline"]["cores"] == 1
        assert o["cpu_baseline"]["loglik_rel_diff_vs_gpu"] < 1e-9
    assert abs(two["loglik"] - one["loglik"]) <= 1e-12 * abs(one["loglik"])
    assert two["allreduce_plus_copy_ms"] > 0 and two["config"]["allreduces_per_step"] == 1.0
    assert two["statistics_identical_on_all_ranks"] is True
    assert one["configs1_gaussian"]["roofline"]["alg_bytes_per_timestep"] == 144
    assert one["configs1_gaussian"]["cpu_baseline"]["loglik_rel_diff_vs_gpu"] < 1e-9
    assert one["configs3_64_states"]["self_checks_fired"] == 0
    assert all(g["self_checks_fired"] == 0 and g["tile_kernels"] for g in one["more_than_64_states"])
    for o in (one, two):
        whole = [e for e in o["secondary"] if "WHOLE" in e["config"]]
        assert len(whole) == 5 and all(e["n_gpus"] == o["n_gpus"] for e in whole)
        assert all(e.get("ms_per_iteration", e.get("ms_per_sweep")) > 0 for e in whole)
    # the EM sequence of the sharded run is the single-process sequence
    em = [[e for e in o["secondary"] if "WHOLE EM" in e["config"]] for o in (one, two)]
    for a, b in zip(*em):
        np.testing.assert_allclose(a["loglik_first_last"], b["loglik_first_last"], rtol=1e-12)


@pytest.mark.gpu
def test_bench_collective_path_on_rccl_with_one_rank(launcher):
    """The driver launches `bench.py --gpus N` under torch.distributed.run with backend nccl (= RCCL).  A
    one-GPU box can run that code path with ONE rank: RANK / WORLD_SIZE in the environment make the bench take
    its distributed branch -- statistics into a device buffer, `all_reduce` on the engine's stream, one
    non-blocking copy, the MIN / MAX comparison of the ranks' statistics, the max-over-ranks timing -- on RCCL,
    and the line must equal the plain one-process run's."""
    small = ["--steps", "2", "--warmup", "1", "--ntraj", "16", "--length", "30000", "--cpu-traj", "2",
             "--no-secondary", "--no-steady"]
    plain = _bench_line(launcher, 1, small)
    import json
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small
    r = launcher.run([cmd], timeout=900, env=env)[0]
    assert r["rc"] == 0, r["out"]
    line = [ln for ln in r["out"].splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, r["out"]
    rccl = json.loads(line[0])
    assert rccl["config"]["collective"] == "RCCL all-reduce" and rccl["config"]["allreduces_per_step"] == 1.0
    assert rccl["statistics_identical_on_all_ranks"] is True and rccl["allreduce_plus_copy_ms"] > 0
    assert abs(rccl["loglik"] - plain["loglik"]) <= 1e-13 * abs(plain["loglik"])
    assert rccl["cpu_baseline"]["loglik_rel_diff_vs_gpu"] < 1e-9
    assert rccl["roofline"]["alg_bytes_per_launch"] == plain["roofline"]["alg_bytes_per_launch"]


@pytest.mark.gpu
def test_bench_eight_rank_dry_run(launcher):
    """The 8-GPU job of BASELINE configs[2] before an 8-GPU node exists: `bench.py --gpus 8
    --oversubscribe` (eight ranks sharing this box's GPU, the sum over gloo -- everything else is the
    production path): 1024 trajectories in 128-trajectory shards, rank r drawing trajectories
    r*128 .. by global index, ONE all-reduce per step, identical reduced statistics on every rank,
    and the log-likelihood of the one-rank run on the same (short) trajectories.  No scaling number
    is expected from it."""
    small = ["--steps", "2", "--warmup", "1", "--ntraj", "1024", "--length", "3000", "--no-cpu",
             "--no-secondary", "--no-steady"]
    one = _bench_line(launcher, 1, small)
    eight = _bench_line(launcher, 8, small, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    assert eight["config"]["trajectories_per_gpu"] == 128 and eight["config"]["trajectories_total"] == 1024
    assert eight["config"]["allreduces_per_step"] == 1.0
    assert eight["statistics_identical_on_all_ranks"] is True
    assert abs(eight["loglik"] - one["loglik"]) <= 1e-12 * abs(one["loglik"])
    assert eight["roofline"]["alg_bytes_per_launch"] == 136 * 128 * 3000
    assert abs(eight["value"] - 1024 * 3000 * 2 / (eight["ms_per_step"] * 2e-3)) < 1e-6 * eight["value"]


@pytest.mark.gpu
def test_allreduce_of_the_statistics_through_the_c_abi_on_rccl():
    """include/bhmm_amd.h section 2b: bhmm_comm_init_rank + bhmm_ctx_allreduce_stats (RCCL loaded by the
    library itself, no torch.distributed).  One rank is what a one-GPU box can run: the statistics an
    E-step left in a device buffer go through ncclAllReduce on the engine's stream and come back
    unchanged; the communicator reports its geometry; a buffer on another device is refused."""
    import torch
    from bhmm_amd.engine import Engine, NativeComm
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    n = 8
    A = rng.random((n, n)) + 0.1
    A /= A.sum(axis=1)[:, None]
    pi = np.full(n, 1.0 / n)
    mu, sig = np.linspace(-4, 4, n), np.linspace(0.6, 1.5, n)
    obs = [rng.normal(0, 3, T) for T in (5000, 1234, 77)]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, n)
    stats = torch.zeros(eng.stats_size, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    comm = NativeComm(0)                                   # nranks = 1
    assert (comm.nranks, comm.rank) == (1, 0)
    eng.estep_launch(A, pi, mu, sig, stats_dev=stats.data_ptr())
    comm.allreduce_stats(eng, stats.data_ptr(), eng.stats_size)
    eng.sync()
    res = eng.unpack(stats.cpu().numpy())
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    np.testing.assert_allclose(res.loglik, ref["logL"].sum(), rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9)
    comm.close()
    eng.close()
