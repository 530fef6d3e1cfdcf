#!/usr/bin/env python
"""bench.py -- E-step throughput of the MI355X forward-backward engine.

Metric (BASELINE.json): timesteps/s of one full E-step (fused emission probabilities +
scaled forward + backward + gamma / xi / emission sufficient statistics), whole job.
One "step" of this benchmark = one E-step over the whole resident batch, from "model handed
to the engine" to "reduced statistics on the host".

Workload at N GPUs: BASELINE.json configs[1] per GPU -- 8-state Gaussian HMM,
256 trajectories x 1e5 time steps of synthetic observations (weak scaling: every rank holds
its own 256 trajectories; the packed sufficient statistics are all-reduced over RCCL).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NSTATES = 8
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
B_ALG_GAUSS = 2 * 8 + 16 * NSTATES   # 144 B / time step, SURVEY.md section 8(d)


# ---------------------------------------------------------------------------------------
# synthetic workload (recipe of bhmm/util/testsystems.py:26-65,159-160 restated)
# ---------------------------------------------------------------------------------------
def metastable_matrix(n, rng, lifetime_min=10.0, lifetime_max=100.0):
    lt = np.linspace(np.log(lifetime_min), np.log(lifetime_max), n)
    diag = 1.0 - 1.0 / np.exp(lt)
    X = rng.random((n, n))
    X = X + X.T
    T = X / X.sum(axis=1)[:, None]
    for i in range(n):
        T[i, i] = 0.0
        T[i, :] *= (1.0 - diag[i]) / T[i, :].sum()
        T[i, i] = 1.0 - T[i, :].sum()
    return T


def stationary(T):
    p = np.full(T.shape[0], 1.0 / T.shape[0])
    for _ in range(20000):
        q = p @ T
        if np.abs(q - p).max() < 1e-15:
            break
        p = q
    return p / p.sum()


def make_c2_model(n=NSTATES, seed=2):
    """Generating model + the (perturbed) model the E-step is evaluated with."""
    rng = np.random.default_rng(1000 * seed)
    A = metastable_matrix(n, rng)
    mu = np.linspace(-5.0, 5.0, n)
    sigma = np.linspace(0.5, 2.0, n)
    pi = stationary(A)
    return dict(A=A, mu=mu, sigma=sigma, pi=pi, A_eval=0.9 * A + 0.1 / n, mu_eval=mu + 0.1)


def synth_hidden(model, K, T, seed):
    """K hidden paths of length T, vectorised over trajectories."""
    rng = np.random.default_rng(seed)
    A, pi = model["A"], model["pi"]
    n = A.shape[0]
    cdf = np.cumsum(A, axis=1)
    cdf[:, -1] = 1.0
    s = np.empty((T, K), dtype=np.int8)
    cur = np.minimum(np.searchsorted(np.cumsum(pi), rng.random(K)), n - 1)
    s[0] = cur
    for t in range(1, T):
        u = rng.random(K)
        cur = (u[:, None] > cdf[cur]).sum(axis=1)
        s[t] = cur
    return np.ascontiguousarray(s.T)


def synth_gaussian(model, K, T, seed):
    s = synth_hidden(model, K, T, seed)
    rng = np.random.default_rng(seed + 7919)
    return model["mu"][s] + model["sigma"][s] * rng.standard_normal((K, T))


def synth_gaussian_device(model, K, T, seed, device):
    """Observations as one trajectory-concatenated fp64 tensor on `device`."""
    import torch
    obs = synth_gaussian(model, K, T, seed)
    return torch.from_numpy(obs.reshape(-1)).to(device)


# ---------------------------------------------------------------------------------------
# CPU baseline: the reference's own C kernels (oracle/_ref, compiled from the reference
# sources) driven through the call sequence of maximum_likelihood.py:249-265, one core.
# Falls back to this repo's restatement (oracle/liboracle.so, kind "port").
# ---------------------------------------------------------------------------------------
def cpu_baseline(model, obs_sample):
    from oracle import oracle as orc
    A = np.ascontiguousarray(model["A_eval"])
    pi = np.ascontiguousarray(model["pi"])
    mu = np.ascontiguousarray(model["mu_eval"])
    sig = np.ascontiguousarray(model["sigma"])
    K, T = obs_sample.shape
    n = A.shape[0]
    use_ref = orc.ref_available()
    pobs = np.zeros((T, n))
    alpha = np.zeros((T, n))
    beta = np.zeros((T, n))
    gamma = np.zeros((T, n))
    C = np.zeros((n, n))
    ll = 0.0
    t0 = time.perf_counter()
    for k in range(K):
        o = np.ascontiguousarray(obs_sample[k])
        if use_ref:
            orc.ref_pobs_gaussian(o, mu, sig, out=pobs)
            outl = np.where(pobs.sum(axis=1) == 0)[0]      # outputmodel.py:126-130
            if outl.size:
                pobs[outl, :] = 1.0
            l, _ = orc.ref_forward(A, pobs, pi, alpha)
            orc.ref_backward(A, pobs, beta)
            orc.ref_gamma(alpha, beta, out=gamma)           # hidden/api.py:176-186
            orc.ref_transition_counts(alpha, beta, A, pobs, C)
        else:
            r = orc.estep("gaussian", [o], A, pi, mu, sig)
            l = r["logL"][0]
        ll += l
    dt = time.perf_counter() - t0
    return dict(value=K * T / dt, unit="timesteps/s", cores=1,
                kind="reference" if use_ref else "port",
                sample="%d of the workload's trajectories x %d steps, 8-state Gaussian, "
                       "p_obs+forward+backward+gamma+xi per trajectory as in "
                       "maximum_likelihood.py:249-265, %.1f s on one core" % (K, T, dt)), ll


# ---------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ntraj", type=int, default=256)
    ap.add_argument("--length", type=int, default=100000)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--cpu-traj", type=int, default=200, help="trajectories in the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bhmm_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (even with one rank): bring up RCCL
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    K, T = args.ntraj, args.length
    model = make_c2_model()
    obs_host = synth_gaussian(model, K, T, seed=1000 * 2 + rank)   # this rank's trajectories
    obs_dev = torch.from_numpy(obs_host.reshape(-1)).to(dev)
    off = np.arange(K + 1, dtype=np.int64) * T

    stream = torch.cuda.current_stream(dev)
    eng = Engine(local, stream=stream.cuda_stream)
    eng.set_observations_device("gaussian", obs_dev.data_ptr(), off, NSTATES, chunk=args.chunk)
    S = eng.stats_size
    stats = torch.zeros(S, dtype=torch.float64, device=dev)
    host_stats = torch.zeros(S, dtype=torch.float64).pin_memory()

    def one_step():
        if not distributed:
            # single GPU: the library lands the statistics in pinned host memory itself
            eng.estep_launch(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
            host_stats.copy_(torch.from_numpy(eng.estep_fetch().packed))
            return host_stats
        eng.estep_launch(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"],
                         stats_dev=stats.data_ptr())
        dist.all_reduce(stats)                          # RCCL sum of the packed statistics
        host_stats.copy_(stats, non_blocking=True)
        stream.synchronize()                            # statistics are on the host
        return host_stats

    def fence():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    kern_ms = np.zeros(5)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        eng.sync()
        kern_ms += [eng.kernel_ms(i) for i in range(5)]
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    res = eng.unpack(host_stats.numpy().copy())
    assert np.isfinite(res.loglik)
    # sanity of the reduced statistics: every step carries unit gamma mass
    np.testing.assert_allclose(res.state_counts.sum(), world * K * T, rtol=1e-9)

    if rank == 0:
        steps_total = world * K * T
        value = steps_total * args.steps / elapsed
        kern_ms /= args.steps
        names = ["prescan", "stitch", "fwdbwd", "finalize", "estep_total"]
        dom = int(np.argmax(kern_ms[:4]))
        # algorithmic bytes of the canonical two-pass algorithm: the streaming
        # kernel k_estep carries them (obs twice, alpha written once and read once).
        alg_bytes_launch = B_ALG_GAUSS * K * T
        achieved = alg_bytes_launch / (kern_ms[2] * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01", "r01q_traffic.json")
        if os.path.exists(tj) and (K, T) == (256, 100000):
            # HBM bytes of the two sweep launches of one E-step from the PMC counters (collected offline with
            # rocprofv3, separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied)
            if eng.get_option("spec_ok") > 0 and eng.get_option("spec_fail") == 0:
                traffic = json.load(open(tj))["traffic_bytes_per_launch"]
        out = {
            "metric": "timesteps/sec forward-backward (whole node), N=8 states",
            "value": value, "unit": "timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 8-state Gaussian HMM, %d trajectories x %d "
                                   "timesteps per GPU, one full E-step" % (K, T),
                       "trajectories_per_gpu": K, "timesteps_per_trajectory": T,
                       "chunk_len": eng.chunk_len, "chunks": eng.num_chunks,
                       "speculative_boundaries": {k: eng.get_option(k) for k in
                                                  ("spec_enabled", "spec_W", "spec_ok", "spec_fail",
                                                   "spec_last_dev")},
                       "parallelism": "trajectories sharded over %d GPU(s), RCCL all-reduce of "
                                      "%d statistics" % (world, S)},
            "roofline": {"bound": "hbm", "kernel": "k_estep_light<8,gauss,spec,P1> + k_estep<8,gauss,spec,P2> (the sweep of one E-step)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC, profiles/r01/r01q_traffic.json)",
                         "alg_bytes_per_launch": alg_bytes_launch,
                         "alg_bytes_per_timestep": B_ALG_GAUSS,
                         "whole_estep_frac": B_ALG_GAUSS * value / world / 1e9 / HBM_PEAK_GBS},
            "kernel_ms": {names[i]: float(kern_ms[i]) for i in range(5)},
            "dominant_kernel": names[dom],
        }
        if world == 1 and not args.no_cpu:
            cb, ll_cpu = cpu_baseline(model, obs_host[: args.cpu_traj])
            # parity on the very same trajectories, asserted in the same run
            eng2 = Engine(local, stream=stream.cuda_stream)
            sub = obs_dev[: args.cpu_traj * T]
            eng2.set_observations_device("gaussian", sub.data_ptr(),
                                         off[: args.cpu_traj + 1], NSTATES)
            r2 = eng2.estep(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
            rel = abs(r2.loglik - ll_cpu) / abs(ll_cpu)
            assert rel < 1e-9, "GPU/CPU log-likelihood mismatch %g" % rel
            cb["loglik_rel_diff_vs_gpu"] = rel
            eng2.close()
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
