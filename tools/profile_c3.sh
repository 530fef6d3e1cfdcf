#!/bin/bash
# tools/profile_c3.sh TAG: the configs[3] block of bench.py (E-step, Viterbi, Gibbs path step) alone, then the same
# command under rocprofv3 --kernel-trace --stats: TAG_c3.json, TAG_c3_kernel_stats.csv (gpurun_out/).
tag=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --only c3 > $O/${tag}_c3.json 2> /tmp/c3.err || tail -5 /tmp/c3.err
rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 $R/bench.py --only c3 > $O/${tag}_c3_under_rocprof.json 2> /tmp/prof_c3.err
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $O/${tag}_c3_kernel_stats.csv
python3 - $O/${tag}_c3.json $O/${tag}_c3_kernel_stats.csv <<'PY'
import json, sys, csv
d = json.load(open(sys.argv[1]))["configs3_64_states"]
print("E-step %.3f ms | Viterbi %s | Gibbs %s" % (d["ms"], json.dumps(d["viterbi"]), json.dumps(d["gibbs_path_step"])))
for r in csv.DictReader(open(sys.argv[2])):
    if float(r["Percentage"]) > 0.3:
        print("%-90s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
