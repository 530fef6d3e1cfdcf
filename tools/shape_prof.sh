#!/bin/bash
# tools/shape_prof.sh n K T: kernel statistics of tools/shape_scan.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pss
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pss -- python3 $R/tools/shape_scan.py "$@" 2>&1 | grep "^n="
python3 - $(find /tmp/pss -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-96s calls %5s  avg %9.1f us" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
