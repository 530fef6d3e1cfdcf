#!/bin/bash
# tools/c4_prof.sh TAG: configs[3] (64 states) E-step time + kernel statistics of tools/c4_once.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-c4}
python3 $R/tools/c4_once.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_time.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc4 -- python3 $R/tools/c4_once.py > /dev/null 2>&1
cp $(find /tmp/pc4 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
python3 - $O/${T}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-100s calls %5s  avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
