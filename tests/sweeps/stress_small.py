"""Test infrastructure (uses the oracle).  Random parity sweep against the oracle (E-step statistics, Viterbi, sampled paths) over small random
shapes: 1..24 states, gaussian / discrete (alphabets on both sides of the LDS limits), ragged
trajectories, default and odd chunk lengths; gaussian data also with far outliers and very narrow
states.  Where some observation has all its densities in the denormal range (< 2.3e-308) the
reference's own row sums are denormals of a few bits: there the results are
compared with the reference's recursions carried out in 80-bit arithmetic on the reference's own
double-precision emission rows; everywhere else with the oracle itself, to 1e-10 / 1e-8.
Prints one line per failure and a summary; exit code 1 on failure."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd import _lib
if os.environ.get("BHMM_AMD_LIB"):
    _lib.SIGNATURES.pop("bhmm_diag_gauss_pdf", None)
from bhmm_amd.engine import Engine
from oracle import oracle as orc
from ld_reference import estep_longdouble

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = limited = needles = 0
for case in range(ncase):
    n = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 25))
    kind = "gaussian" if rng.random() < 0.5 else "discrete"
    K = int(rng.integers(1, 7))
    lens = [int(x) for x in rng.integers(1, int(rng.choice([40, 400, 3000, 12000])), K)]
    if os.environ.get("LARGE"):  # few, long trajectories: default chunk plans, the branch-free kernels
        K = int(rng.integers(1, 40))
        lens = [int(x) for x in rng.integers(20000, 200000, K)]
    chunk = int(rng.choice([0, 0, 1, 7, 16, 33, 100]))
    A = rng.random((n, n)) + rng.choice([0.0, 2.0, 10.0]) * np.eye(n)
    if n > 1 and rng.random() < 0.3:  # sparse transitions (every row keeps its diagonal)
        mask = rng.random((n, n)) < 0.5
        np.fill_diagonal(mask, True)
        A = A * mask + 1e-300 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    if n > 1 and rng.random() < 0.2:
        pi[rng.integers(0, n)] = 0.0
        pi /= pi.sum()
    if kind == "gaussian":
        regime = rng.choice(["plain", "far", "narrow", "needle"], p=[0.3, 0.3, 0.3, 0.1])
        mu, sig = np.sort(rng.normal(0, 3, n)), rng.uniform(0.3, 2.0, n)
        if regime == "narrow":
            sig = sig * 0.05
        obs = [rng.normal(0, 12.0 if regime == "far" else 3.0, T) for T in lens]
        if regime == "needle":
            # densities of 1e30 .. 1e75: widths far below the spacing of the means, observations drawn from a
            # hidden path of the model itself (anything else would be an outlier row in every step)
            sig = sig * float(rng.choice([1e-30, 1e-60, 1e-75]))
            if n > 1 and rng.random() < 0.5:   # ... for some of the states only: rows with entries of 1e40 beside
                wide = rng.random(n) < 0.5     # ordinary and denormal ones
                sig = np.where(wide, rng.uniform(0.3, 30.0, n), sig)
            obs = []
            for T in lens:
                st = np.zeros(T, dtype=np.int64)
                st[0] = rng.integers(0, n)
                for t in range(1, T):
                    nzs = np.flatnonzero(A[st[t - 1]] > 0)
                    st[t] = rng.choice(nzs) if rng.random() < 0.3 else st[t - 1] if A[st[t - 1], st[t - 1]] > 0 else rng.choice(nzs)
                obs.append(mu[st] + sig[st] * rng.normal(0, 1, T))
        if rng.random() < 0.15:  # a few extreme values: huge, infinite, exactly on a mean
            for o in obs:
                for _ in range(int(rng.integers(1, 4))):
                    o[rng.integers(0, len(o))] = rng.choice([1e200, -1e200, np.inf, -np.inf, 1e-300, float(mu[rng.integers(0, n)])])
        par = (mu, sig)
        M = 0
    else:
        M = int(rng.choice([2, 17, 64, 300, 1150, 1300, 4000]))
        B = rng.dirichlet(np.ones(M) * 0.5, size=n) * 0.98 + 0.02 / M
        if rng.random() < 0.2:  # emission probabilities spread over hundreds of decades
            B = np.exp(-float(rng.choice([50.0, 300.0, 700.0])) * rng.random((n, M)))
            B /= B.sum(axis=1, keepdims=True)
        if rng.random() < 0.3:  # symbols some states cannot emit (never a whole column)
            zero = rng.random((n, M)) < 0.3
            zero[rng.integers(0, n, M), np.arange(M)] = False
            B = np.where(zero, 0.0, B)
            B /= B.sum(axis=1, keepdims=True)
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lens]
        par = (B, None)
    tag = "case %d: %s n=%d M=%d K=%d lens=%s chunk=%d" % (case, kind, n, M, K, lens, chunk)
    try:
        with np.errstate(all="ignore"):
            ref = orc.estep(kind, obs, A, pi, *par)
        if not (np.all(np.isfinite(ref["logL"])) and np.all(np.isfinite(ref["C"]))):
            continue  # the reference itself leaves the floating-point range here
        needles += int(kind == "gaussian" and regime == "needle")
        eng = Engine(0)
        eng.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        denorm = False
        if kind == "gaussian":
            # emission ENTRIES in the denormal range, not only whole rows: with a sparse transition matrix
            # one such entry can be the only way on, and the reference's own products with it keep a few
            # bits (seed 8101 case 681: its log-likelihood is 2.3e-5 off the 80-bit recursion, which the
            # kernels match)
            with np.errstate(all="ignore"):
                for o in obs:
                    pe = orc.pobs_gaussian(o, *par)
                    denorm |= bool(np.any((pe < 2.3e-308) & (pe > 0)))
        else:
            denorm = bool(np.any((B < 2.3e-308) & (B > 0)))   # (the same for emission matrices)
        for rep in range(2):
            res = eng.estep(A, pi, *par)
            if denorm:
                if rep == 0:
                    ld_logL, ld_C = estep_longdouble(A, pi, [orc.pobs_gaussian(o, *par) if kind == "gaussian" else orc.pobs_discrete(o, B) for o in obs])
                ok = np.allclose(res.logL_k, ld_logL, rtol=1e-10, atol=1e-9) and np.allclose(res.C, ld_C, rtol=1e-8, atol=1e-9)
                if not ok:
                    # where the reference itself is far off the 80-bit recursion (seed 8001 case 278: 3.5e-3
                    # in the log-likelihood, rows of a few bits), being as close to it as the reference is
                    # -- here the kernels follow the reference to 3.6e-6 -- is all that can be asked
                    # (seed 16001 case 2087 -- a step in which every state that carries alpha has p = 0 and the
                    # step's likelihood is a denormal of three bits, 3.4585e-323: the reference is 3.5 % off in
                    # that factor; the kernels re-form such products from the row times 2^900 since round 4;
                    # tests/golden/cases/gauss7_denormal_entries_16001_2087.npz)
                    ok = (np.abs(res.logL_k - ld_logL).max() <= max(2.0 * np.abs(ref["logL"] - ld_logL).max(), 1e-9) and
                          np.abs(res.C - ld_C).max() <= max(10.0 * np.abs(ref["C"] - ld_C).max(), 1e-9))
                if not ok:
                    print("  (denormal regime; vs 80-bit recursion) logL", np.abs(res.logL_k - ld_logL).max(), "C", np.abs(res.C - ld_C).max(),
                          "| reference vs 80-bit:", np.abs(ref["logL"] - ld_logL).max(), np.abs(ref["C"] - ld_C).max())
            else:
                ok = np.allclose(res.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and np.allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
            if not ok:
                bad += 1
                print("ESTEP MISMATCH", tag, "rep", rep, np.abs(res.logL_k - ref["logL"]).max(), np.abs(res.C - ref["C"]).max())
                if os.environ.get("SAVE") and rep == 0:
                    np.savez(os.path.join(os.environ["SAVE"], "stress_case_%d_%d.npz" % (int(sys.argv[1]) if len(sys.argv) > 1 else 7, case)),
                             A=A, pi=pi, par0=par[0], par1=par[1] if par[1] is not None else np.zeros(0), kind=kind, chunk=chunk,
                             obs=np.concatenate(obs), lens=np.array(lens))
                if os.environ.get("VERBOSE"):
                    print("  gpu logL", res.logL_k, "ref", ref["logL"], "gpu C", res.C, "ref C", ref["C"], "par", par,
                          "careful", eng.get_option("careful"), "spec ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"),
                          "W", eng.get_option("spec_W"), "nan obs", [int(np.isnan(o).sum()) for o in obs])
        pobs = [orc.pobs_gaussian(o, *par) if kind == "gaussian" else orc.pobs_discrete(o, B) for o in obs]
        if not denorm and not os.environ.get("LARGE") and case % 3 == 0:
            # gamma rows of a stored-gamma E-step, and the same E-step through explicit emission rows
            res_g = eng.estep(A, pi, *par, store_gamma=True)
            kk = int(case // 3) % len(obs)
            al, be = orc.forward(A, pobs[kk], pi)[1], orc.backward(A, pobs[kk])
            with np.errstate(all="ignore"):
                g_ref = orc.gamma(al, be)
            g_ok = not np.all(np.isfinite(g_ref)) or np.allclose(eng.gamma(kk), g_ref, rtol=1e-8, atol=1e-12)
            if not g_ok:
                # ill-conditioned rows (seed 10201 case 2301: an absorbing state and emission probabilities
                # 130 decades apart -- the reference's own gamma is 2e-11 off the 80-bit recursion in one
                # entry): no further from the 80-bit rows than ten times the reference's own distance
                from ld_reference import hidden_longdouble
                with np.errstate(all="ignore"):
                    g_ld = np.asarray(hidden_longdouble(A, pobs[kk], pi)[3], dtype=np.float64)
                d_ref, d_gpu = np.abs(g_ref - g_ld), np.abs(eng.gamma(kk) - g_ld)
                g_ok = bool(np.all(np.isfinite(g_ld))) and bool(np.all(d_gpu <= np.maximum(10.0 * d_ref.max(), 1e-12 + 1e-8 * np.abs(g_ld))))
                if g_ok and d_ref.max() <= 1e-12:
                    g_ok = False   # the reference is accurate here: the deviation is the kernels' own
                if not g_ok:
                    print("  (vs 80-bit rows) gpu", d_gpu.max(), "reference", d_ref.max())
            if not g_ok:
                bad += 1
                print("GAMMA MISMATCH", tag, "traj", kk, np.abs(eng.gamma(kk) - g_ref).max())
                if os.environ.get("SAVE"):
                    np.savez(os.path.join(os.environ["SAVE"], "stress_case_%d_%d.npz" % (int(sys.argv[1]) if len(sys.argv) > 1 else 7, case)),
                             A=A, pi=pi, par0=par[0], par1=par[1] if par[1] is not None else np.zeros(0), kind=kind, chunk=chunk,
                             obs=np.concatenate(obs), lens=np.array(lens))
            if not np.allclose(res_g.logL_k, ref["logL"], rtol=1e-10, atol=1e-10):
                bad += 1
                print("STORE-GAMMA ESTEP MISMATCH", tag)
            e2 = Engine(0)
            e2.set_observations("explicit", pobs, n, chunk=chunk)
            r2 = e2.estep(A, pi, None, None)
            if not (np.allclose(r2.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and np.allclose(r2.C, ref["C"], rtol=1e-8, atol=1e-10)):
                bad += 1
                print("EXPLICIT ESTEP MISMATCH", tag, np.abs(r2.logL_k - ref["logL"]).max(), np.abs(r2.C - ref["C"]).max())
                if os.environ.get("SAVE"):
                    np.savez(os.path.join(os.environ["SAVE"], "explicit_case_%d_%d.npz" % (int(sys.argv[1]) if len(sys.argv) > 1 else 7, case)),
                             A=A, pi=pi, chunk=chunk, pobs=np.concatenate(pobs), lens=np.array(lens))
            e2.close()
        vp = eng.viterbi(A, pi, *par)
        for k, (p, po) in enumerate(zip(vp, pobs)):
            vr = orc.viterbi(A, po, pi)
            if not np.array_equal(p, vr):
                bad += 1
                print("VITERBI MISMATCH", tag, "traj", k, "differing steps", int((p != vr).sum()), "of", len(vr), "(denormal regime)" if denorm else "")
        if denorm:
            eng.close()
            continue  # (sampled paths: the reference's forward rows are degenerate there)
        u = [rng.random(T) for T in lens]
        sp = eng.sample_paths(A, pi, *par, u=u)[0]
        for k, (p, po, uu) in enumerate(zip(sp, pobs, u)):
            if not np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu)):
                bad += 1
                print("SAMPLE MISMATCH", tag, "traj", k)
        eng.close()
    except Exception as e:  # noqa
        if "is not finite" in repr(e) and "statistic" in repr(e):
            # the loud failure of a known limitation: a REDUCIBLE transition matrix (blocks that never
            # mix) whose blocks' likelihood ratio leaves the double range within one chunk
            from scipy.sparse.csgraph import connected_components
            if connected_components(A > 0, directed=True, connection="strong")[0] > 1:
                limited += 1
                try:
                    eng.close()
                except Exception:
                    pass
                continue
        if os.environ.get("SAVE"):
            np.savez(os.path.join(os.environ["SAVE"], "stress_case_%d_%d.npz" % (int(sys.argv[1]) if len(sys.argv) > 1 else 7, case)),
                     A=A, pi=pi, par0=par[0], par1=par[1] if par[1] is not None else np.zeros(0), kind=kind, chunk=chunk,
                     obs=np.concatenate(obs), lens=np.array(lens))
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300])
print("stress: %d cases, %d failures%s%s" % (ncase, bad, (" (%d reducible models refused loudly: non-finite statistics)" % limited) if limited else "",
                                             (" (%d with densities of 1e30 and more)" % needles) if needles else ""))
sys.exit(1 if bad else 0)
