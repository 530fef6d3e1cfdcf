#!/bin/bash
# tools/profile_r06.sh TAG: round-6 evidence for profiles/r06 (run on the GPU box through gpurun).
#   everything tools/profile_r05.sh writes (bench line, kernel statistics of exactly the timed launches,
#   FETCH / WRITE / SQ_INSTS_VALU passes -> TAG_traffic.json, TAG_clock.txt), plus
#   TAG_c3.json, TAG_c3_kernel_stats.csv        the configs[3] block of bench.py alone and under rocprofv3
#                                               (tools/profile_c3.sh): E-step, Viterbi, Gibbs path step
#   TAG_shard.json, TAG_shard_kernel_stats.csv  rank 0's shard of the 8-GPU projection (128 x 1e6, 1-rank RCCL)
tag=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
bash $R/tools/profile_r05.sh $tag
bash $R/tools/profile_c3.sh $tag
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --only shard --steps 20 --warmup 5 > $O/${tag}_shard.json 2> /tmp/sh.err || tail -5 /tmp/sh.err
rm -rf /tmp/prof_sh
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -- python3 $R/bench.py --only shard --steps 20 --warmup 5 > /tmp/sh_prof.json 2> /tmp/prof_sh.err
cp $(find /tmp/prof_sh -name "*kernel_stats.csv" | head -1) $O/${tag}_shard_kernel_stats.csv
python3 - $O/${tag}_shard.json $O/${tag}_shard_kernel_stats.csv <<'PY'
import json, sys, csv
d = json.loads(open(sys.argv[1]).read().split("\n")[0])["projected_8gpu"]
print({k: d[k] for k in ("shard", "shard_ms", "shard_kernels_ms", "allreduce_plus_copy_ms", "host_gap_ms", "roofline_frac_of_the_shard")})
for r in csv.DictReader(open(sys.argv[2])):
    if float(r["Percentage"]) > 0.5:
        print("%-90s calls %5s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
