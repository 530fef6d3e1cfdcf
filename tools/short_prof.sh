#!/bin/bash
# tools/short_prof.sh K T: kernel statistics of tools/short_once.py
R=$GRAFT_REPO_ROOT
python3 $R/tools/short_once.py $1 $2 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/psh
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psh -- python3 $R/tools/short_once.py $1 $2 > /dev/null 2>&1
python3 - $(find /tmp/psh -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-100s calls %5s  avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
