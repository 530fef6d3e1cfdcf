import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian_device
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = synth_gaussian_device(model, K, T, seed=2000, device="cuda:0")
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
ms = []
for i in range(140):
    try:
        eng.estep_launch(*args); eng.sync()
    except Exception as e:
        pass
    ms.append(eng.kernel_ms(2))
print(os.environ.get("BHMM_AMD_LIB", "default").split("_")[-1], "sweep ms median %.4f min %.4f" % (np.median(ms[70:]), min(ms[70:])), "spec_fail", eng.get_option("spec_fail"))
