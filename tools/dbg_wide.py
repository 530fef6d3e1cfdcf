import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
n = int(sys.argv[1]); K = int(sys.argv[2]); T = int(sys.argv[3])
rng = np.random.default_rng(n)
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 1.0, n)
g = torch.Generator(device=dev); g.manual_seed(n)
s = torch.randint(0, n, (K, T // 50), device=dev, generator=g).repeat_interleave(50, dim=1)
obs = (torch.tensor(mu, device=dev)[s] + torch.tensor(sig, device=dev)[s]
       * torch.randn((K, T), device=dev, dtype=torch.float64, generator=g)).reshape(-1)
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
print("set ok segs", eng.get_option("wide_segments"), flush=True)
args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
for it in range(3):
    r = eng.estep(*args)
    torch.cuda.synchronize()
    print("estep", it, "W", eng.get_option("spec_W"), "segs", eng.get_option("wide_segments"), "ok/fail",
          eng.get_option("spec_ok"), eng.get_option("spec_fail"), "dev", eng.get_option("spec_last_dev"), flush=True)
