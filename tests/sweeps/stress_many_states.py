"""Test infrastructure (uses the oracle).  Random parity sweep for the two kernel families that the small-
shape sweep does not reach: 9..64 states (lane-per-state kernels, wide_kernels.hpp) and 65..200 states
(any-N family: tile_kernels.hpp / gen_kernels.hpp; MANY_MAXN=420 MANY_PBIG=0.8: mostly 65..420 states, i.e.
also the matrix-core kernels of big_kernels.hpp) -- gaussian / discrete, ragged trajectories (lengths 1, 2, ... included),
sparse transition matrices, zero entries in pi, far-away observations.  E-step statistics against the
oracle (logL 1e-10, counts 1e-8), Viterbi and sampled paths (given the uniforms) bit for bit.
Prints one line per failure and a summary; exit code 1 on failure."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
from ld_reference import estep_longdouble

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ONLY = int(os.environ["ONLY"]) if os.environ.get("ONLY") else -1   # replay one case (same random stream)
MAXN = int(os.environ.get("MANY_MAXN", 200))   # (MANY_MAXN=420: also the 129 .. 512-state kernels of big_kernels.hpp)
PBIG = float(os.environ.get("MANY_PBIG", 0.35))
bad = skipped = nbig = 0
for case in range(ncase):
    big = rng.random() < PBIG
    n = int(rng.integers(65, MAXN + 1)) if big else int(rng.integers(9, 65))
    kind = "gaussian" if rng.random() < 0.5 else "discrete"
    K = int(rng.integers(1, 6))
    tmax = int(rng.choice([30, 300, 700])) if big else int(rng.choice([40, 500, 4000]))
    if big and os.environ.get("MANY_TBIG"):   # (long trajectories: the segment-parallel Viterbi passes above 64 states)
        tmax = int(os.environ["MANY_TBIG"])
    lens = [int(x) for x in rng.integers(1, tmax, K)]
    chunk = int(rng.choice([0, 0, 0, 17, 64, 250]))
    A = rng.random((n, n)) ** 2 + rng.choice([0.0, 3.0, 20.0]) * np.eye(n)
    if rng.random() < 0.4:
        mask = rng.random((n, n)) < 0.4
        np.fill_diagonal(mask, True)
        A = A * mask
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    if rng.random() < 0.3:
        pi[rng.integers(0, n, 3)] = 0.0
        pi /= pi.sum()
    if kind == "gaussian":
        mu, sig = np.sort(rng.normal(0, 0.15 * n, n)), rng.uniform(0.3, 2.0, n)
        spread = 0.2 * n * (4.0 if rng.random() < 0.3 else 1.0)
        obs = [rng.normal(0, spread, T) for T in lens]
        if rng.random() < 0.15:  # a few extreme values: huge, infinite, exactly on a mean
            for o in obs:
                for _ in range(int(rng.integers(1, 4))):
                    o[rng.integers(0, len(o))] = rng.choice([1e200, -1e200, np.inf, -np.inf, 1e-300, float(mu[rng.integers(0, n)])])
        par, M = (mu, sig), 0
    else:
        M = int(rng.choice([3, 40, 257, 1200]))
        B = rng.dirichlet(np.ones(M) * 0.5, size=n) * 0.98 + 0.02 / M
        if rng.random() < 0.25:  # emission probabilities spread over hundreds of decades
            B = np.exp(-float(rng.choice([50.0, 300.0, 700.0])) * rng.random((n, M)))
            B /= B.sum(axis=1, keepdims=True)
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lens]
        par = (B, None)
    tag = "case %d: %s n=%d M=%d lens=%s chunk=%d" % (case, kind, n, M, lens, chunk)
    try:
        with np.errstate(all="ignore"):
            ref = orc.estep(kind, obs, A, pi, *par)
        if not (np.all(np.isfinite(ref["logL"])) and np.all(np.isfinite(ref["C"]))):
            skipped += 1
            continue
        pobs = [orc.pobs_gaussian(o, *par) if kind == "gaussian" else orc.pobs_discrete(o, B) for o in obs]
        # emission rows whose entries reach into the denormal range: the double-precision reference loses
        # digits there (seed 3007 case 894: its counts are off the 80-bit recursion by 1.5e-6, the kernels
        # by 1e-14) -- compared with the reference's recursions carried out in 80-bit arithmetic instead
        denorm = kind == "gaussian" and any(np.any((po.max(axis=1) < 1e-250)) for po in pobs)
        if denorm:
            with np.errstate(all="ignore"):
                ld_logL, ld_C = estep_longdouble(A, pi, pobs)
            if not (np.all(np.isfinite(ld_logL)) and np.all(np.isfinite(ld_C))):
                skipped += 1
                continue
        nbig += int(big)
        if ONLY >= 0 and case != ONLY:
            for T in lens:
                rng.random(T)
            continue
        if ONLY >= 0 and os.environ.get("SAVE"):
            np.savez(os.path.join(os.environ["SAVE"], "many_states_case_%s_%d.npz" % (sys.argv[1], case)), A=A, pi=pi,
                     par0=par[0], par1=par[1] if par[1] is not None else np.zeros(0), kind=kind, chunk=chunk,
                     obs=np.concatenate(obs), lens=np.array(lens), M=M)
        eng = Engine(0)
        eng.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        for rep in range(2):
            res = eng.estep(A, pi, *par)
            ok = False
            if denorm:
                # The lane-per-state kernels stay with the 80-bit values (1e-14 where the reference is off
                # by 1e-6); the any-N family is order-faithful -- its rows and log-likelihoods are the
                # reference's bit for bit -- but forms the counts as a GEMM over W = p o beta / S, and
                # where those products are denormal neither it nor the reference keeps its digits (the
                # reference is off the 80-bit counts by up to 7e-2 in this sweep): there the counts only
                # have to be as good as the reference's own, or within 1e-5.
                ok = (np.allclose(res.logL_k, ld_logL, rtol=1e-10, atol=1e-9) and
                      np.allclose(res.C, ld_C, rtol=1e-8, atol=1e-9))
                if not ok and big:
                    ok = (np.allclose(res.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and
                          np.abs(res.C - ld_C).max() <= max(10.0 * np.abs(ref["C"] - ld_C).max(), 1e-5))
            if not ok:
                ok = (np.allclose(res.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and
                      np.allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10) and
                      np.allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=1e-8, atol=1e-12) and
                      np.allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10))
            if not ok:
                bad += 1
                print("ESTEP MISMATCH", tag, "rep", rep, np.abs(res.logL_k - ref["logL"]).max(), np.abs(res.C - ref["C"]).max(),
                      ("| denormal regime: C gpu-80bit %.2e, ref-80bit %.2e" % (np.abs(res.C - ld_C).max(), np.abs(ref["C"] - ld_C).max())) if denorm else "")
                if ONLY >= 0:
                    d = np.abs(res.C - ref["C"])
                    i, j = np.unravel_index(np.argmax(d), d.shape)
                    print("   worst C entry", (i, j), "gpu", res.C[i, j], "ref", ref["C"][i, j], "| C sums", res.C.sum(), ref["C"].sum(),
                          "| gamma0", np.abs(res.gamma0_sum - ref["gamma0_sum"]).max(), "state_counts", np.abs(res.state_counts - ref["state_counts"]).max(),
                          "| careful", eng.get_option("careful"), "spec ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"),
                          "segments", eng.get_option("wide_segments"))
        if denorm:
            eng.close()
            for T in lens:
                rng.random(T)
            continue  # (paths: the reference's own rows are degenerate there, stress_small)
        if case % 3 == 0:
            # stored gamma rows, and the same E-step through explicit emission rows
            res_g = eng.estep(A, pi, *par, store_gamma=True)
            kk = int(case // 3) % len(obs)
            g_ref = orc.gamma(orc.forward(A, pobs[kk], pi)[1], orc.backward(A, pobs[kk]))
            if np.all(np.isfinite(g_ref)) and not np.allclose(eng.gamma(kk), g_ref, rtol=1e-8, atol=1e-12):
                bad += 1
                print("GAMMA MISMATCH", tag, "traj", kk, np.abs(eng.gamma(kk) - g_ref).max())
            if not np.allclose(res_g.logL_k, ref["logL"], rtol=1e-10, atol=1e-10):
                bad += 1
                print("STORE-GAMMA ESTEP MISMATCH", tag)
            e2 = Engine(0)
            e2.set_observations("explicit", pobs, n, chunk=chunk)
            r2 = e2.estep(A, pi, None, None)
            if not (np.allclose(r2.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and np.allclose(r2.C, ref["C"], rtol=1e-8, atol=1e-10)):
                bad += 1
                print("EXPLICIT ESTEP MISMATCH", tag, np.abs(r2.logL_k - ref["logL"]).max(), np.abs(r2.C - ref["C"]).max())
            e2.close()
        if os.environ.get("MANY_VITW"):
            eng.set_option("viterbi_W", int(os.environ["MANY_VITW"]))
        vp = eng.viterbi(A, pi, *par)
        for k, (p, po) in enumerate(zip(vp, pobs)):
            vr = orc.viterbi(A, po, pi)
            if not np.array_equal(p, vr):
                bad += 1
                print("VITERBI MISMATCH", tag, "traj", k, "differing steps", int((p != vr).sum()), "of", len(vr))
        u = [rng.random(T) for T in lens]
        sp = eng.sample_paths(A, pi, *par, u=u)[0]
        for k, (p, po, uu) in enumerate(zip(sp, pobs, u)):
            if not np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu)):
                bad += 1
                print("SAMPLE MISMATCH", tag, "traj", k)
        eng.close()
    except Exception as e:  # noqa
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300])
print("stress_many_states: %d cases (%d skipped: the reference itself leaves the floating-point range; %d with more than 64 states), %d failures" % (ncase, skipped, nbig, bad))
sys.exit(1 if bad else 0)
