"""Test infrastructure (uses the oracle).  Debug helper for a saved explicit-emission case of stress_small.py."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
d = np.load(sys.argv[1])
A, pi, pobs, lens = d["A"], d["pi"], d["pobs"], d["lens"]
n = A.shape[0]
po = np.split(pobs, np.cumsum(lens)[:-1])
print("A", A, "pi", pi)
for chunk in (int(d["chunk"]), 0):
    for k, p in enumerate(po):
        eng = Engine(0)
        eng.set_observations("explicit", [p], n, chunk=chunk)
        r = eng.estep(A, pi, None, None, store_gamma=True)
        al, be = orc.forward(A, p, pi)[1], orc.backward(A, p)
        Cr = orc.transition_counts(al, be, A, p)
        g = eng.gamma(0)
        print("chunk", chunk, "traj", k, "T", len(p), "logL", r.loglik, "C nan", int(np.isnan(r.C).sum()), "max|dC|", np.nanmax(np.abs(r.C - Cr)),
              "gamma nan rows", np.where(np.isnan(g).any(axis=1))[0][:6], "careful", eng.get_option("careful"), "chunks", eng.num_chunks)
        r2 = eng.estep(A, pi, None, None)
        print("     without gamma: C nan", int(np.isnan(r2.C).sum()), "SG", r2.state_counts)
        if np.isnan(r2.C).any():
            for t in range(len(p)):
                print("      t", t, "pobs", p[t], "alpha", al[t], "beta", be[t])
        eng.close()
