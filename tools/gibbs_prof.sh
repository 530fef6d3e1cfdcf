#!/bin/bash
# kernel statistics of the Gibbs path step at K trajectories x 1e5 steps (tools/gibbs_parts.py K)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -- python3 $R/tools/gibbs_parts.py $1 > /tmp/g.out 2> /tmp/g.err
grep "K=" /tmp/g.out
python3 - $(find /tmp/prof_g -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print("%-100s calls %6s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
