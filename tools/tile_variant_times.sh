#!/bin/bash
# kernel times of the configs[3] E-step (tools/c3_estep_time.py under rocprofv3) for the settings given as arguments,
# e.g.  bash tools/tile_variant_times.sh "BHMM_AMD_TILE_PSTORE=0" "BHMM_AMD_TILE_PSTORE=1"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pk
  export $v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/tools/c3_estep_time.py 2 > /tmp/o.txt 2>&1
  tail -1 /tmp/o.txt | cut -c1-120
  python3 - "$v" <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True)[0])):
    if "k_tile" in r["Name"] and int(r["Calls"]) > 3:
        print("  %s  %s calls %s avg %.1f us" % (sys.argv[1], r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
