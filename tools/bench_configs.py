#!/usr/bin/env python
"""Secondary measurements on the other BASELINE.json configs (not the bench.py line):
configs[2] (8-state discrete, 1024 x 1e6, on ONE GPU), configs[3] (64-state Gaussian, 128 x 1e5),
configs[4] (Gibbs hidden-path sweep, 8 states, 256 x 1e5), Viterbi on configs[1].
Prints one JSON object per measurement."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_c2_model, metastable_matrix, stationary  # noqa: E402
from bhmm_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)


def hidden_paths_gpu(A, pi, K, T, seed):
    """K hidden paths of length T sampled on the GPU (time-serial, trajectory-parallel)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cdf = torch.tensor(np.cumsum(A, axis=1), device=dev)
    cdf[:, -1] = 1.0
    cur = torch.searchsorted(torch.tensor(np.cumsum(pi), device=dev),
                             torch.rand(K, device=dev, dtype=torch.float64, generator=g)).clamp_(max=len(pi) - 1)
    out = torch.empty((T, K), dtype=torch.int8, device=dev)
    out[0] = cur
    u = torch.rand((T, K), device=dev, dtype=torch.float64, generator=g)
    for t in range(1, T):
        cur = (u[t].unsqueeze(1) > cdf[cur]).sum(dim=1)
        out[t] = cur
    return out.t().contiguous()


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    which = sys.argv[1:] or ["c2v", "c5", "c4", "c3"]
    res = []
    if "c2v" in which or "c5" in which:
        model = make_c2_model()
        K, T = 256, 100000
        s = hidden_paths_gpu(model["A"], model["pi"], K, T, 11).long()
        obs = (torch.tensor(model["mu"], device=dev)[s] + torch.tensor(model["sigma"], device=dev)[s]
               * torch.randn((K, T), device=dev, dtype=torch.float64)).reshape(-1)
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
        args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
        if "c2v" in which:
            dt = timeit(lambda: eng.viterbi(*args), 3)
            res.append(dict(config="configs[1] Viterbi, 8-state Gaussian 256 x 1e5", seconds=dt,
                            timesteps_per_s=K * T / dt, note="includes copying 102 MB of paths to the host",
                            chunked=eng.get_option("viterbi_chunked"),
                            close_decisions=eng.get_option("viterbi_close"),
                            boundary_dev=eng.get_option("spec_last_dev")))
        if "c5" in which:
            dt = timeit(lambda: eng.sample_paths(*args, seed=1, want_paths=False), 5)
            res.append(dict(config="configs[4] Gibbs hidden-path sweep (forward + backward sampling + "
                                   "path statistics), 8-state Gaussian 256 x 1e5, one GPU",
                            seconds=dt, timesteps_per_s=K * T / dt, sweeps_100_seconds=100 * dt))
        eng.close()
        del obs, s
    if "c4" in which:
        rng = np.random.default_rng(64)
        n, K, T = 64, 128, 100000
        A = metastable_matrix(n, rng)
        pi = stationary(A)
        mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
        s = hidden_paths_gpu(A, pi, K, T, 12).long()
        obs = (torch.tensor(mu, device=dev)[s] + torch.tensor(sig, device=dev)[s]
               * torch.randn((K, T), device=dev, dtype=torch.float64)).reshape(-1)
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
        for _ in range(4):          # lets the segment plan settle on a warm-up length that verifies
            eng.estep(*args)
        dt = timeit(lambda: eng.estep(*args), 3)
        r = eng.estep(*args)
        assert abs(r.state_counts.sum() - K * T) < 1e-6 * K * T
        res.append(dict(config="configs[3] E-step, 64-state Gaussian 128 x 1e5 (wide family)",
                        seconds=dt, timesteps_per_s=K * T / dt,
                        segments=eng.get_option("wide_segments"),
                        spec={k: eng.get_option(k) for k in ("spec_W", "spec_ok", "spec_fail",
                                                             "spec_last_dev")}))
        eng.close()
        del obs, s
    if "c3" in which:
        rng = np.random.default_rng(3000)
        n, M, K, T = 8, 64, 1024, 1000000
        A = metastable_matrix(n, rng)
        pi = stationary(A)
        B = rng.dirichlet(np.ones(M), size=n)
        obs = torch.empty(K * T, dtype=torch.int32, device=dev)
        cdfB = torch.tensor(np.cumsum(B, axis=1), device=dev)
        cdfB[:, -1] = 1.0
        KB = 128                                  # generate in blocks of trajectories
        for b in range(0, K, KB):
            s = hidden_paths_gpu(A, pi, KB, T, 100 + b).long()
            u = torch.rand((KB, T), device=dev, dtype=torch.float64)
            o = (u.unsqueeze(2) > cdfB[s]).sum(dim=2).to(torch.int32)
            obs[b * T:(b + KB) * T] = o.reshape(-1)
            del s, u, o
        eng = Engine(0)
        eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T,
                                    n, nsymbols=M)
        args = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
        dt = timeit(lambda: eng.estep(*args), 3)
        r = eng.estep(*args)
        assert abs(r.state_counts.sum() - K * T) < 1e-6 * K * T
        res.append(dict(config="configs[2] E-step, 8-state discrete (M=64) 1024 x 1e6 on ONE GPU",
                        seconds=dt, timesteps_per_s=K * T / dt, chunk_len=eng.chunk_len,
                        chunks=eng.num_chunks,
                        kernel_ms={k: eng.kernel_ms(i) for i, k in
                                   enumerate(["prescan", "stitch", "fwdbwd", "finalize", "total"])},
                        hbm_alg_GBs=136 * K * T / dt / 1e9))
        eng.close()
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
