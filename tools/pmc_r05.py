"""Workload of the round-5 PMC passes (tools/profile_r05.sh): three E-steps of the bench line's two
workloads with the bench's own data and models -- configs[2] (8-state discrete, 1024 x 1e6, the
headline) and configs[1] (8-state Gaussian, 256 x 1e5) -- plus the 1 GiB calibration copy that
MI355X_MICROARCH.md (section HBM) prescribes for FETCH_SIZE / WRITE_SIZE."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from bench import Ranks, Series, workload_configs1, workload_configs2

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rk = Ranks(torch, dist, 1, 0, 0, dev, "nccl", False, stream)
shapes = {"configs2": (1024, 1000000), "configs1": (256, 100000)}
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
if len(argv) > 1:                           # a smaller configs[2] for a dry run: K T
    shapes["configs2"] = (int(argv[0]), int(argv[1]))
wls = [workload_configs2(*shapes["configs2"])]
if "--only-configs2" not in sys.argv:       # (bench.py's live passes: the headline workload and the calibration copy)
    wls.append(workload_configs1(*shapes["configs1"]))
for wl in wls:
    ser = Series(rk, wl, wl.K, 0)
    for _ in range(4):                      # the first calibrates the warm-up; the last three are read
        ser.one_step()
    print(wl.key, "spec_W", ser.eng.get_option("spec_W"), "spec_fail", ser.eng.get_option("spec_fail"),
          "chunks", ser.eng.num_chunks, "chunk_len", ser.eng.chunk_len)
    ser.close()
x = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()   # 1 GiB
for _ in range(3):
    y = x.clone()                                                     # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("done")
