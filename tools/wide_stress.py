"""Stress of the 9..64-state segmented kernels against the serial plan on the same GPU:
random state counts, ragged lengths, random segment lengths / warm-ups, both kinds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhmm_amd.engine import Engine
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(9, 65))
    kind = "gaussian" if seed % 3 else "discrete"
    M = int(rng.integers(5, 40))
    A = rng.random((n, n)) + 0.02
    A[rng.random((n, n)) < 0.2] = 0.0
    A += np.eye(n) * rng.uniform(0.2, 3.0)
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    lengths = [int(x) for x in rng.integers(1, 60, 3)] + [int(x) for x in rng.integers(400, 5000, 4)]
    if kind == "gaussian":
        p0, p1 = np.linspace(-6, 6, n), rng.uniform(0.3, 1.2, n)
        obs = [rng.normal(0, 4, T) for T in lengths]
    else:
        p0, p1 = rng.dirichlet(np.ones(M), n), None
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    kw = dict(nsymbols=M) if kind == "discrete" else {}
    ser = Engine(0)
    ser.set_option("wide_segments", 0)
    ser.set_observations(kind, obs, n, **kw)
    rs = ser.estep(A, pi, p0, p1)
    seg = Engine(0)
    seg.set_option("wide_segment_len", int(rng.integers(50, 700)))
    if seed % 2:
        seg.set_option("spec_W", int(rng.integers(100, 300)))
    seg.set_observations(kind, obs, n, **kw)
    rg = seg.estep(A, pi, p0, p1)
    rg2 = seg.estep(A, pi, p0, p1)
    scale = np.abs(rs.packed).max()
    err = np.abs(rg2.packed - rs.packed).max() / scale
    rel_ll = np.abs(rg2.logL_k - rs.logL_k).max() / np.abs(rs.logL_k).max()
    ok = err < 1e-9 and rel_ll < 1e-11
    bad += not ok
    print(seed, kind, "n", n, "segs", seg.get_option("wide_segments"), "W", seg.get_option("spec_W"),
          "ok/fail", seg.get_option("spec_ok"), seg.get_option("spec_fail"), "careful", seg.get_option("careful"),
          "err %.2e ll %.2e" % (err, rel_ll), "" if ok else "  <-- MISMATCH")
    ser.close(); seg.close()
print("mismatches:", bad)
