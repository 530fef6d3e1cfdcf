"""Test infrastructure (uses the oracle).  Debug helper for a saved case of stress_small.py (gaussian / discrete):
E-step on the GPU under the options that select the kernels, against the oracle."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
d = np.load(sys.argv[1], allow_pickle=True)
kind = str(d["kind"])
A, pi, lens, chunk = d["A"], d["pi"], d["lens"], int(d["chunk"])
par0 = d["par0"]; par1 = d["par1"] if d["par1"].size else None
obs = np.split(d["obs"], np.cumsum(lens)[:-1])
if kind == "discrete":
    obs = [o.astype(np.int32) for o in obs]
n = A.shape[0]; M = par0.shape[1] if kind == "discrete" else 0
print(kind, "n", n, "M", M, "lens", lens, "chunk", chunk)
if kind == "discrete":
    print("B min positive %.3e, column maxima min %.3e" % (par0[par0 > 0].min(), par0.max(axis=0).min()))
ref = orc.estep(kind, obs, A, pi, par0, par1)
for spec in (1, 0):
    for sg in (False, True):
        eng = Engine(0)
        eng.set_option("spec_enabled", spec)
        eng.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        r = eng.estep(A, pi, par0, par1, store_gamma=sg)
        print("spec", spec, "store_gamma", sg, ": logL diff %.2e" % np.abs(r.logL_k - ref["logL"]).max(), "C nan", int(np.isnan(r.C).sum()),
              "max|dC| %s" % np.nanmax(np.abs(r.C - ref["C"])), "SG nan", int(np.isnan(r.state_counts).sum()),
              "careful", eng.get_option("careful"), "ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"))
        eng.close()
