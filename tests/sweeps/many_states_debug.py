"""Test infrastructure (uses the oracle).  Debug helper: E-step of a saved case of stress_many_states.py, per
trajectory, against the oracle: which entries of the count matrix deviate.
python tests/sweeps/many_states_debug.py case.npz"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
d = np.load(sys.argv[1], allow_pickle=True)
A, pi, mu, sig, lens = d["A"], d["pi"], d["par0"], d["par1"], d["lens"]
obs = np.split(d["obs"], np.cumsum(lens)[:-1])
n = A.shape[0]
print("n", n, "lens", lens, "A zeros", int((A == 0).sum()), "sig", sig.min(), sig.max())
for k, o in enumerate(obs):
    ref = orc.estep("gaussian", [o], A, pi, mu, sig)
    eng = Engine(0)
    eng.set_observations("gaussian", [o], n)
    res = eng.estep(A, pi, mu, sig)
    dC = np.abs(res.C - ref["C"])
    i, j = np.unravel_index(np.argmax(dC), dC.shape)
    rel = dC / np.maximum(np.abs(ref["C"]), 1e-300)
    print("traj", k, "T", len(o), "logL diff", abs(res.logL_k[0] - ref["logL"][0]), "max |dC|", dC.max(), "at", (i, j), "gpu", res.C[i, j], "ref", ref["C"][i, j],
          "| entries with rel > 1e-8 and |C| > 1e-6:", int(((rel > 1e-8) & (ref["C"] > 1e-6)).sum()), "careful", eng.get_option("careful"))
    if dC.max() > 1e-9:
        po = orc.pobs_gaussian(o, mu, sig)
        ll, al = orc.forward(A, po, pi)
        be = orc.backward(A, po)
        # per-step xi of the worst entry
        for t in range(len(o) - 1):
            x = al[t][:, None] * A * (po[t + 1] * be[t + 1])[None, :]
            x /= x.sum()
            print("   t", t, "xi[%d,%d] %.6e" % (i, j, x[i, j]), "max pobs next", po[t + 1].max(), "argmax gamma", int(np.argmax(al[t] * be[t])))
    eng.close()
from ld_reference import estep_longdouble
pobs = [orc.pobs_gaussian(o, mu, sig) for o in obs]
with np.errstate(all="ignore"):
    ld_logL, ld_C = estep_longdouble(A, pi, pobs)
ref = orc.estep("gaussian", list(obs), A, pi, mu, sig)
eng = Engine(0)
eng.set_observations("gaussian", list(obs), n)
res = eng.estep(A, pi, mu, sig)
print("whole batch: max |C_gpu - C_80bit| %.3e   max |C_ref - C_80bit| %.3e   max |C_gpu - C_ref| %.3e" % (
    np.abs(res.C - ld_C).max(), np.abs(ref["C"] - ld_C).max(), np.abs(res.C - ref["C"]).max()))
print("smallest positive emission entry %.3e, smallest row maximum %.3e" % (
    min(p[p > 0].min() for p in pobs), min(p.max(axis=1).min() for p in pobs)))
