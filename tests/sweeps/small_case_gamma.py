"""Test infrastructure (uses the oracle).  Stored gamma rows of one trajectory of a saved stress_small case on the
GPU against the reference's alpha / beta / gamma around the first non-finite row."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from bhmm_amd import _lib
from oracle import oracle as orc
d = np.load(sys.argv[1], allow_pickle=True)
kind = str(d["kind"]); A, pi, lens = d["A"], d["pi"], d["lens"]
par0 = d["par0"]; par1 = d["par1"] if d["par1"].size else None
o = np.split(d["obs"], np.cumsum(lens)[:-1])[int(sys.argv[2])]
if kind == "discrete":
    o = o.astype(np.int32)
n = A.shape[0]; M = par0.shape[1] if kind == "discrete" else 0
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else int(d["chunk"])
po = orc.pobs_discrete(o, par0) if kind == "discrete" else orc.pobs_gaussian(o, par0, par1)
al = orc.forward(A, po, pi)[1]; be = orc.backward(A, po); gr = orc.gamma(al, be)
eng = Engine(0)
eng.set_observations(kind, [o], n, nsymbols=M, chunk=chunk)
L = eng._L
packed = np.empty(eng.stats_size); 
rc = L.bhmm_estep(eng._h, _lib.dp(_lib.f64(A)), _lib.dp(_lib.f64(pi)), _lib.dp(_lib.f64(par0)), None if par1 is None else _lib.dp(_lib.f64(par1)), None, _lib.FLAG_STORE_GAMMA)
print("estep rc", rc, "chunks", eng.num_chunks, "x", eng.chunk_len, "careful", eng.get_option("careful"), "spec ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"))
g = eng.gamma(0)
bad = np.where(~np.isfinite(g).all(axis=1))[0]
print("non-finite gamma rows:", bad[:10], "... total", len(bad), "of", len(o))
if len(bad):
    for t in range(max(0, bad[0] - 3), min(len(o), bad[0] + 4)):
        print(" t", t, "gpu gamma", g[t], "ref gamma", gr[t], "ref alpha", al[t], "ref beta", be[t], "p", po[t])
