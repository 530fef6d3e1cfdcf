"""Test infrastructure (uses the oracle).  Shrinks trajectory K of a saved stress_small case to a short stretch
whose E-step counts are NaN on the GPU although the reference's are finite; prints it."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
d = np.load(sys.argv[1], allow_pickle=True)
kind = str(d["kind"])
A, pi, lens = d["A"], d["pi"], d["lens"]
par0 = d["par0"]; par1 = d["par1"] if d["par1"].size else None
obs = np.split(d["obs"], np.cumsum(lens)[:-1])[int(sys.argv[2])]
if kind == "discrete":
    obs = obs.astype(np.int32)
n = A.shape[0]; M = par0.shape[1] if kind == "discrete" else 0
def run(o, knd=kind):
    eng = Engine(0)
    if knd == "explicit":
        po = orc.pobs_discrete(o, par0) if kind == "discrete" else orc.pobs_gaussian(o, par0, par1)
        eng.set_observations("explicit", [po], n)
        r = eng.estep(A, pi, None, None)
    else:
        eng.set_observations(kind, [o], n, nsymbols=M)
        r = eng.estep(A, pi, par0, par1)
    eng.close()
    return r
def bad(o):
    if len(o) < 2:
        return False
    ref = orc.estep(kind, [o], A, pi, par0, par1)
    return bool(np.isnan(run(o).C).any() and np.all(np.isfinite(ref["C"])))
assert bad(obs)
print("explicit rows instead: C nan", int(np.isnan(run(obs, "explicit").C).sum()))
a, b = 0, len(obs)
ch = True
while ch:
    ch = False
    if b - a > 2 and bad(obs[a + 1:b]):
        a += 1; ch = True
    if b - a > 2 and bad(obs[a:b - 1]):
        b -= 1; ch = True
o = obs[a:b]
print("minimal stretch", a, b, "obs", o.tolist())
po = orc.pobs_discrete(o, par0) if kind == "discrete" else orc.pobs_gaussian(o, par0, par1)
al = orc.forward(A, po, pi)[1]; be = orc.backward(A, po)
for t in range(len(o)):
    print("  t", t, "p", po[t], "alpha", al[t], "beta", be[t])
print("A", A, "pi", pi)
