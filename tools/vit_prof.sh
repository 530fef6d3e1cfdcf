#!/bin/bash
# tools/vit_prof.sh TAG: Viterbi timing + kernel statistics of tools/viterbi_time.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-vit}
python3 $R/tools/viterbi_time.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_time.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 $R/tools/viterbi_time.py > /dev/null 2>&1
cp $(find /tmp/pv -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
python3 - $O/${T}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-90s calls %5s  avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
