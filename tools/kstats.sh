#!/bin/bash
# tools/kstats.sh OUT script.py [args]: rocprofv3 --kernel-trace --stats of one tool, the per-kernel
# table into gpurun_out/OUT (run on the GPU box through gpurun)
out=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/"$@" > /tmp/ks.log 2>&1
tail -4 /tmp/ks.log
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$out
python3 - $R/gpurun_out/$out <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-100s calls %6s avg %10.1f us  total %9.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
