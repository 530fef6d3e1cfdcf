import os
import sys; sys.path.insert(0,os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))) + ''); sys.path.insert(0,os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))) + '/tests')
import numpy as np
from test_viterbi_margin_gpu import _model,_data
from oracle import oracle as orc
from bhmm_amd.engine import Engine
for n,kind in [(72,"gaussian"),(96,"discrete"),(128,"gaussian"),(64,"gaussian"),(40,"discrete")]:
    rng=np.random.default_rng((7100 if n>64 else 7300)+n); M=21 if n>64 else 17
    A,pi,p0,p1=_model(n,rng,kind,M)
    lengths=(24001,1,9000,2,700) if n>64 else (30011,1,9000,257)
    obs,pobs=_data(kind,rng,lengths,n,M,p0,p1)
    eng=Engine(0); eng.set_option("viterbi_seg_per_simd",1)
    eng.set_observations(kind,obs,n,nsymbols=M if kind=="discrete" else 0)
    if n<=64: eng.set_option("viterbi_margin",2)
    for W in (0,48,200):
        if W: eng.set_option("viterbi_W",W)
        eng.viterbi(A,pi,p0,p1)
        print(n,kind,"W",eng.get_option("viterbi_W"),"segs",eng.get_option("viterbi_segments"),"mism",eng.get_option("viterbi_mismatch"),"far",eng.get_option("viterbi_far"),"used",eng.get_option("viterbi_margin_used"),"close",eng.get_option("viterbi_margin_close"),"rounds",eng.get_option("viterbi_rounds"))
    eng.close()
