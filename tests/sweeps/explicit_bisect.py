"""Test infrastructure (uses the oracle).  Shrinks a saved explicit-emission trajectory to a short one whose
stored gamma rows are NaN on the GPU although the reference's are finite."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
d = np.load(sys.argv[1])
A, pi, pobs, lens = d["A"], d["pi"], d["pobs"], d["lens"]
n = A.shape[0]
p = np.split(pobs, np.cumsum(lens)[:-1])[int(sys.argv[2])]
def bad(q):
    if len(q) < 2:
        return False
    eng = Engine(0)
    eng.set_observations("explicit", [q], n)
    r = eng.estep(A, pi, None, None)
    eng.close()
    al, be = orc.forward(A, q, pi)[1], orc.backward(A, q)
    with np.errstate(all="ignore"):
        Cr = orc.transition_counts(al, be, A, q)
    return bool(np.isnan(r.C).any() and np.all(np.isfinite(Cr)))
assert bad(p)
a, b = 0, len(p)
changed = True
while changed:
    changed = False
    if b - a > 2 and bad(p[a + 1:b]):
        a += 1; changed = True
    if b - a > 2 and bad(p[a:b - 1]):
        b -= 1; changed = True
q = p[a:b]
print("minimal stretch rows", a, b, "of", len(p))
for row in q:
    print("   ", repr(row.tolist()))
print("A", repr(A.tolist()), "pi", repr(pi.tolist()))
al, be = orc.forward(A, q, pi)[1], orc.backward(A, q)
print("ref alpha", al, "ref beta", be)
eng = Engine(0)
eng.set_observations("explicit", [q], n)
r = eng.estep(A, pi, None, None, store_gamma=True)
print("gpu gamma", eng.gamma(0), "gpu C", r.C, "logL", r.loglik, "ref logL", orc.forward(A, q, pi)[0])
