"""Test infrastructure.  Random sweep of E-step SEQUENCES on drifting models: an engine with the carried
boundary vectors (DESIGN.md section 3, A') against one without, every step: packed statistics, per-trajectory
log-likelihoods; drift patterns incl. sudden jumps, repeated models, alternating models, shrinking and growing
drift; 2..8 states, gaussian / discrete, ragged lengths, chunk lengths 256..4096; the last step also against
the oracle.  usage: python tests/sweeps/stress_carry.py [seed [cases]]"""
import os, sys, warnings
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = used_total = fails_total = steps_total = 0
for case in range(ncase):
    n = int(rng.integers(2, 9))
    kind = "gaussian" if rng.random() < 0.6 else "discrete"
    K = int(rng.integers(1, 12))
    lens = [int(x) for x in rng.integers(1, int(rng.choice([20000, 120000])), K)]
    lens[0] = max(lens[0], 30000)
    chunk = int(rng.choice([256, 512, 1024, 2048, 4096]))
    stick = float(rng.choice([2.0, 6.0, 20.0]))
    A0 = rng.random((n, n)) + stick * np.eye(n); A0 /= A0.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        p0, p1 = np.sort(rng.normal(0, 3, n)), rng.uniform(0.5, 1.5, n)
        obs = [rng.normal(0, 3, T) for T in lens]
        M = 0
    else:
        M = int(rng.choice([4, 30, 200]))
        p0, p1 = rng.dirichlet(np.ones(M) * 0.5, size=n), None
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lens]
    tag = "case %d: %s n=%d K=%d chunk=%d stick=%g" % (case, kind, n, K, chunk, stick)
    try:
        a, b = Engine(0), Engine(0)
        a.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        b.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        b.set_option("carry", 0)
        pattern = rng.choice(["shrink", "grow", "jumps", "repeat", "alternate"])
        models = []
        nsteps = int(rng.integers(6, 16))
        for it in range(nsteps):
            if pattern == "shrink":
                d = 10 ** rng.uniform(-3, -2) * 0.6 ** it
            elif pattern == "grow":
                d = 1e-6 * 3.0 ** it
            elif pattern == "jumps":
                d = 1e-4 if it % 4 else 0.2
            elif pattern == "repeat":
                d = 0.0 if it % 2 else 1e-4
            else:
                d = 1e-3
            if pattern == "alternate" and it >= 2:
                models.append(models[it - 2])
                continue
            base = models[-1] if models else (A0, p0, p1)
            A = base[0] * (1 + d * rng.normal(size=(n, n))); A = np.abs(A); A /= A.sum(axis=1, keepdims=True)
            if kind == "gaussian":
                q0 = base[1] + d * rng.normal(size=n)
                q1 = np.abs(base[2] * (1 + min(d, 0.3) * rng.normal(size=n))) + 1e-3
            else:
                q0 = np.abs(base[1] * (1 + d * rng.normal(size=(n, M)))); q0 /= q0.sum(axis=1, keepdims=True)
                q1 = None
            models.append((A, q0, q1))
        for it, (A, q0, q1) in enumerate(models):
            ra, rb = a.estep(A, pi, q0, q1), b.estep(A, pi, q0, q1)
            steps_total += 1
            used_total += int(a.get_option("carry_W") > 0)
            if not (np.allclose(ra.packed, rb.packed, rtol=1e-8, atol=1e-8) and np.allclose(ra.logL_k, rb.logL_k, rtol=1e-11)):
                bad += 1
                print("MISMATCH", tag, pattern, "step", it, "carry_W", a.get_option("carry_W"),
                      np.abs(ra.packed - rb.packed).max(), np.abs(ra.logL_k - rb.logL_k).max())
                break
        fails_total += int(a.get_option("carry_fail"))
        k0 = int(np.argmin(lens)) if min(lens) > 1 else 0
        ref = orc.estep(kind, [obs[0], obs[k0]], A, pi, q0, q1)
        if not np.allclose([ra.logL_k[0], ra.logL_k[k0]], ref["logL"], rtol=1e-9):
            bad += 1
            print("ORACLE MISMATCH", tag, pattern, ra.logL_k[0], ref["logL"])
        a.close(); b.close()
    except Exception as e:  # noqa
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300])
print("stress_carry: %d cases, %d E-steps, %d on carried starts, %d repeated after a failed check, %d failures"
      % (ncase, steps_total, used_total, fails_total, bad))
sys.exit(1 if bad else 0)
