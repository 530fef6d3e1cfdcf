"""Test infrastructure (uses the oracle).  Random parity sweep for the segment-parallel Viterbi pass and
backward draw of the 9..64-state family (k_wide_viterbi_seg, k_wide_sample_seg): trajectories long enough to be
cut into many time segments, fast and slowly mixing / sparse transition matrices, zero entries in pi,
gaussian and discrete emissions (also spread over hundreds of decades), far-away observations.  Viterbi
paths and sampled paths (given the uniforms) against the oracle bit for bit, twice per engine (the second
call searches a shorter warm-up).  Prints one line per failure and a summary; exit code 1 on failure."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ONLY = int(os.environ["ONLY"]) if os.environ.get("ONLY") else -1   # replay one case (same random stream)
bad = skipped = nseg_v = nseg_s = fixups = serial_v = 0
for case in range(ncase):
    n = int(rng.integers(9, 65))
    kind = "gaussian" if rng.random() < 0.5 else "discrete"
    K = int(rng.integers(1, 5))
    lens = [int(x) for x in rng.integers(1, int(rng.choice([3000, 12000, 40000])), K)]
    A = rng.random((n, n)) ** 2 + rng.choice([0.0, 3.0, 20.0, 200.0]) * np.eye(n)
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
        obs = [rng.normal(0, 0.2 * n * (4.0 if rng.random() < 0.3 else 1.0), T) for T in lens]
        if rng.random() < 0.2:
            for o in obs:
                for _ in range(int(rng.integers(1, 4))):
                    o[rng.integers(0, len(o))] = rng.choice([1e200, -1e200, float(mu[rng.integers(0, n)])])
        par, M = (mu, sig), 0
        pobs = [orc.pobs_gaussian(o, mu, sig) for o in obs]
    else:
        M = int(rng.choice([3, 40, 257]))
        B = rng.dirichlet(np.ones(M) * 0.5, size=n) * 0.98 + 0.02 / M
        if rng.random() < 0.25:
            B = np.exp(-float(rng.choice([50.0, 300.0])) * rng.random((n, M)))
            B /= B.sum(axis=1, keepdims=True)
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lens]
        par = (B, None)
        pobs = [orc.pobs_discrete(o, B) for o in obs]
    u = [rng.random(T) for T in lens]
    tag = "case %d: %s n=%d M=%d lens=%s" % (case, kind, n, M, lens)
    with np.errstate(all="ignore"):
        alphas = [orc.forward(A, po, pi) for po in pobs]
    # non-finite reference; or emission rows in the denormal range, where the reference's own alpha rows have
    # lost their digits and the kernels (which carry such rows scaled) legitimately draw other states
    # (stress_many_states.py)
    if not all(np.isfinite(a[0]) for a in alphas) or (kind == "gaussian" and any(np.any(po.max(axis=1) < 1e-250) for po in pobs)):
        skipped += 1
        continue
    opt_v, opt_s, with_estep = int(rng.choice([1, 2, 8, 64])), int(rng.choice([1, 4, 16, 64])), rng.random() < 0.5
    if ONLY >= 0 and case != ONLY:
        continue
    vref = [orc.viterbi(A, po, pi) for po in pobs]
    sref = [orc.sample_path(a[1], A, u=uu) for a, uu in zip(alphas, u)]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", opt_v)
    eng.set_option("sample_seg_per_simd", opt_s)
    eng.set_observations(kind, obs, n, nsymbols=M)
    if with_estep:                        # (calibrates the forward warm-up the draw's alpha rows use)
        try:
            eng.estep(A, pi, *par)
        except AssertionError as e:       # non-finite statistics: a failure only if the reference's are finite
            with np.errstate(all="ignore"):
                ref = orc.estep(kind, obs, A, pi, *par)
            if np.all(np.isfinite(ref["C"])) and np.all(np.isfinite(ref["logL"])):
                bad += 1
                print("ESTEP RAISED", tag, e)
            else:
                skipped += 1
            eng.close()
            continue
    for rep in range(2):
        vp = eng.viterbi(A, pi, *par)
        if not all(np.array_equal(a, b) for a, b in zip(vp, vref)):
            bad += 1
            print("VITERBI MISMATCH", tag, "rep", rep, "segmented", eng.get_option("viterbi_chunked"),
                  "segments", eng.get_option("viterbi_segments"), "W", eng.get_option("viterbi_W"),
                  "rounds", eng.get_option("viterbi_rounds"), [int((a != b).sum()) for a, b in zip(vp, vref)])
        if eng.get_option("viterbi_chunked"):
            nseg_v += 1
            fixups += int(eng.get_option("viterbi_rounds") > 0)
        elif eng.get_option("viterbi_segments") > K:
            serial_v += 1
        sp = eng.sample_paths(A, pi, *par, u=u)[0]
        if not all(np.array_equal(a, b) for a, b in zip(sp, sref)):
            bad += 1
            print("SAMPLED PATH MISMATCH", tag, "rep", rep, "segmented", eng.get_option("sample_segmented"),
                  "fwd segmented", eng.get_option("sample_forward_segmented"), "segments", eng.get_option("sample_segments"),
                  "W", eng.get_option("sample_W"), "rounds", eng.get_option("sample_rounds"),
                  [int((a != b).sum()) for a, b in zip(sp, sref)])
            if ONLY >= 0:
                for kk, (a, b) in enumerate(zip(sp, sref)):
                    for t in np.nonzero(a != b)[0]:
                        al = alphas[kk][1]
                        ps = al[t] * A[:, b[t + 1]] if t + 1 < len(b) else al[t].copy()
                        S = 0.0
                        for x in ps:
                            S += x
                        acc = np.cumsum(ps / S)
                        print("   traj", kk, "t", t, "gpu", a[t], "ref", b[t], "next", b[t + 1] if t + 1 < len(b) else -1, "r", repr(u[kk][t]),
                              "acc around", [repr(x) for x in acc[max(0, b[t] - 1):b[t] + 2]], "S", S, "alpha max", al[t].max(), "obs", obs[kk][t])
        nseg_s += int(eng.get_option("sample_segmented"))
    eng.close()
print("stress_wide_paths: %d cases (%d skipped: non-finite reference or emission rows in the denormal range), %d Viterbi calls over time segments (%d with "
      "fix-up rounds, %d handed to the serial kernel), %d draws over time segments, %d failures"
      % (ncase, skipped, nseg_v, fixups, serial_v, nseg_s, bad))
sys.exit(1 if bad else 0)
