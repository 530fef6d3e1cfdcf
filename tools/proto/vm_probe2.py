import os, sys
import os
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from test_viterbi_margin_gpu import _model,_data
from bhmm_amd.engine import Engine
for n,kind in [(129,"gaussian"),(160,"gaussian"),(200,"discrete"),(256,"discrete")]:
    rng=np.random.default_rng(7700+n); M=19
    A,pi,p0,p1=_model(n,rng,kind,M)
    lengths=(9001,1,3000,2,650)
    obs,pobs=_data(kind,rng,lengths,n,M,p0,p1)
    eng=Engine(0)
    eng.set_observations(kind,obs,n,nsymbols=M if kind=="discrete" else 0)
    for W in (0,160,600):
        if W: eng.set_option("viterbi_W",W)
        eng.viterbi(A,pi,p0,p1)
        print(n,kind,"W",eng.get_option("viterbi_W"),"chunked",eng.get_option("viterbi_chunked"),"segs",eng.get_option("viterbi_segments"),"mism",eng.get_option("viterbi_mismatch"),"far",eng.get_option("viterbi_far"),"used",eng.get_option("viterbi_margin_used"),"close",eng.get_option("viterbi_margin_close"))
    eng.close()
