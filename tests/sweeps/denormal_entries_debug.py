"""Test infrastructure (uses the oracle).  A saved Gaussian case of stress_small.py whose emission rows hold single
denormal ENTRIES: GPU counts against the reference's, the 80-bit recursion's, and the reference's with
those entries set to zero."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
from ld_reference import estep_longdouble
d = np.load(sys.argv[1], allow_pickle=True)
A, pi, mu, sig, lens = d["A"], d["pi"], d["par0"], d["par1"], d["lens"]
obs = np.split(d["obs"], np.cumsum(lens)[:-1])
n = A.shape[0]
po = [orc.pobs_gaussian(o, mu, sig) for o in obs]
def ref_on(pl):
    C = np.zeros((n, n)); ll = []
    for p in pl:
        l, al = orc.forward(A, p, pi); be = orc.backward(A, p)
        C += orc.transition_counts(al, be, A, p); ll.append(l)
    return np.array(ll), C
ll_ref, C_ref = ref_on(po)
ll_z, C_z = ref_on([np.where(p < 2.3e-308, 0.0, p) for p in po])
with np.errstate(all="ignore"):
    ll_ld, C_ld = estep_longdouble(A, pi, po)
for kind, data in (("gaussian", obs), ("explicit", po)):
    for chunk in (int(d["chunk"]), 0, 1000):
        eng = Engine(0)
        if kind == "gaussian":
            eng.set_observations(kind, data, n, chunk=chunk); r = eng.estep(A, pi, mu, sig)
        else:
            eng.set_observations(kind, data, n, chunk=chunk); r = eng.estep(A, pi, None, None)
        print(kind, "chunk", chunk, "chunks", eng.num_chunks, "| logL: vs ref %.2e vs 80bit %.2e vs zeroed %.2e | C: vs ref %.2e vs 80bit %.2e vs zeroed %.2e" % (
            np.abs(r.logL_k - ll_ref).max(), np.abs(r.logL_k - ll_ld).max(), np.abs(r.logL_k - ll_z).max(),
            np.abs(r.C - C_ref).max(), np.abs(r.C - C_ld).max(), np.abs(r.C - C_z).max()), "careful", eng.get_option("careful"))
        eng.close()
