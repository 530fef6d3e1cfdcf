#!/bin/bash
# tools/tail_prof.sh: average duration of the tail kernel in the headline loop
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ptl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptl -- python3 $R/bench.py --no-cpu --no-secondary --no-steady > /dev/null 2>&1
python3 - $(find /tmp/ptl -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-80s calls %5s  avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
