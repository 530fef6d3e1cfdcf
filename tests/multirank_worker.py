"""One rank of the multi-process GPU test (tests/test_multirank_gpu.py) -- run as a child process:

    python tests/multirank_worker.py RANK WORLD PORT OUTDIR [BACKEND]

WORLD ranks over gloo, ALL on GPU 0 (the test box has one GPU; RCCL refuses two ranks on one
device, so the collective runs over gloo on the host copy of the device statistics buffer --
everything else is the production path: sharding, device statistics buffer, rank-0 parameter
draws + broadcast).  WORLD == 1 runs the same problem in a single process without a process
group: the reference the sharded runs must reproduce."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def problem():
    rng = np.random.RandomState(11)
    n = 3
    A = np.array([[0.95, 0.04, 0.01], [0.05, 0.9, 0.05], [0.02, 0.08, 0.9]])
    mu, sig = np.array([-2.0, 0.5, 3.0]), np.array([0.6, 0.8, 0.7])
    obs = []
    for T in (6000, 2500, 4100, 1, 3300, 5200, 777):
        s = np.empty(T, dtype=int)
        s[0] = rng.randint(n)
        for t in range(1, T):
            s[t] = rng.choice(n, p=A[s[t - 1]])
        obs.append(mu[s] + sig[s] * rng.randn(T))
    return obs, n


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    import bhmm_amd
    tag = "w%d" % world
    if world > 1 or backend == "nccl":
        # BACKEND nccl with WORLD 1: one rank, but through the RCCL collectives
        # (BHMM_AMD_FORCE_COMM=1): all-reduce / broadcast on the engine's device buffers
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = port
        if backend == "nccl":
            os.environ["BHMM_AMD_FORCE_COMM"] = "1"
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            tag = "rccl%d" % world
        dist.init_process_group(backend, rank=rank, world_size=world)
    obs, n = problem()
    init = bhmm_amd.gaussian_hmm(np.full(n, 1.0 / n), np.full((n, n), 0.1) + 0.7 * np.eye(n),
                                 np.array([-1.0, 0.0, 2.0]), np.ones(n))
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, n, initial_model=init, reversible=False,
                                              accuracy=1e-6, maxit=12, device=0)
    hmm = est.fit()
    from bhmm_amd.engine import Engine
    assert isinstance(est._engine, Engine) and est._engine.device == 0      # the HIP engine
    # rank 0 carries the chain's generator; other ranks are seeded differently on purpose
    np.random.seed(7 if rank == 0 else 500 + rank)
    sampler = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=hmm, reversible=False, device=0)
    chain = sampler.sample(3, save_hidden_state_trajectory=True, seed=3)
    np.savez(os.path.join(outdir, "%s_r%d.npz" % (tag, rank)),
             L=est.likelihoods, A=hmm.transition_matrix, pi=hmm.initial_distribution,
             mu=hmm.output_model.means, sig=hmm.output_model.sigmas, C=est.count_matrix,
             nlocal=len(est.local_trajectories),
             v=np.concatenate(hmm.hidden_state_trajectories),
             chain_A=np.array([m.transition_matrix for m in chain]),
             chain_mu=np.array([m.output_model.means for m in chain]),
             chain_sig=np.array([m.output_model.sigmas for m in chain]),
             chain_paths=np.concatenate(chain[-1].hidden_state_trajectories))
    if world > 1 or backend == "nccl":
        import torch.distributed as dist
        assert est._comm.active and est._comm.backend == backend
        dist.destroy_process_group()
    print("rank %d/%d done" % (rank, world))


if __name__ == "__main__":
    main()
