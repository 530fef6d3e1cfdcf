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


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(launcher):
    """`python bench.py --gpus 2` without a launcher environment starts two ranks itself and
    reports n_gpus = 2 (here both ranks share the box's one GPU: --oversubscribe, gloo)."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = launcher.run([[sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                       "--oversubscribe", "--steps", "3", "--warmup", "1", "--ntraj", "16",
                       "--length", "20000", "--no-cpu"]], timeout=600, env=env)[0]
    assert r["rc"] == 0, r["out"]
    line = [ln for ln in r["out"].splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, r["out"]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 16 * 20000 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
