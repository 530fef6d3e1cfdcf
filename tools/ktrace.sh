#!/bin/bash
# tools/ktrace.sh SCRIPT.py: per-kernel durations (rocprofv3 --kernel-trace --stats) of a python script
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/pk.log 2>&1
tail -2 /tmp/pk.log
f=$(find /tmp/pk -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    print(r['Name'][:48].ljust(48), r['Calls'].rjust(5), r['AverageNs'].rjust(14), r['MinNs'].rjust(10), r['MaxNs'].rjust(10))
" < /dev/null
