"""Test infrastructure (uses the oracle).  Per trajectory of a saved stress_small case: count-matrix and stored-gamma
differences between the GPU and the reference, and the rows around the first gamma row that is off."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from bhmm_amd import _lib
from oracle import oracle as orc
np.set_printoptions(precision=6, linewidth=200)
d = np.load(sys.argv[1], allow_pickle=True)
kind = str(d["kind"]); A, pi, lens = d["A"], d["pi"], d["lens"]
par0 = d["par0"]; par1 = d["par1"] if d["par1"].size else None
obs = np.split(d["obs"], np.cumsum(lens)[:-1])
n = A.shape[0]; M = par0.shape[1] if kind == "discrete" else 0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else int(d["chunk"])
print("A\n", A, "\npi", pi)
for k, o in enumerate(obs):
    if kind == "discrete":
        o = o.astype(np.int32)
    po = orc.pobs_discrete(o, par0) if kind == "discrete" else orc.pobs_gaussian(o, par0, par1)
    ref = orc.estep(kind, [o], A, pi, par0, par1)
    al = orc.forward(A, po, pi)[1]; be = orc.backward(A, po); gr = orc.gamma(al, be)
    eng = Engine(0)
    eng.set_observations(kind, [o], n, nsymbols=M, chunk=chunk)
    st = eng.estep(A, pi, par0, par1, store_gamma=True)
    g = eng.gamma(0)
    dg = np.abs(g - gr).max(axis=1) if len(o) else np.zeros(0)
    from ld_reference import estep_longdouble
    with np.errstate(all="ignore"):
        ld_ll = estep_longdouble(A, pi, [po])[0][0]
    print("   logL gpu %.10f ref %.10f 80-bit %.10f | gpu-80bit %.3e ref-80bit %.3e" % (st.loglik, ref["logL"][0], ld_ll, st.loglik - ld_ll, ref["logL"][0] - ld_ll))
    print("traj", k, "T", len(o), "dlogL %.2e dC %.2e dgamma %.2e careful" % (abs(st.loglik - ref["logL"]), np.abs(st.C - ref["C"]).max() if len(o) > 1 else 0.0, dg.max() if len(o) else 0.0), eng.get_option("careful"))
    bad = np.where(dg > 1e-10)[0]
    if len(bad):
        for t in range(max(0, bad[0] - 2), min(len(o), bad[0] + 3)):
            print("  t", t, "gpu gamma", g[t], "ref gamma", gr[t], "\n     ref alpha", al[t], "beta", be[t], "p", po[t])
    if len(o) > 1 and np.abs(st.C - ref["C"]).max() > 1e-10:
        print("  C gpu\n", st.C, "\n  C ref\n", ref["C"])
