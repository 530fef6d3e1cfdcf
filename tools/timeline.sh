#!/bin/bash
# tools/timeline.sh SCRIPT.py [args]: kernel timeline of a python script (rocprofv3 --kernel-trace):
# for the last 12 E-steps, start offsets / durations / gaps of the kernels between two k_estep_light
# launches.  Output on stdout.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_estep_light' in r[2] and ', 2>' in r[2]]
idx = idx[-13:]
for a, b in zip(idx[:-1], idx[1:]):
    t0 = rows[a][0]
    line = []
    prev_end = None
    for s, e, n in rows[a:b]:
        short = n.split('(')[0].replace('void bhmm::', '').replace('bhmm::', '')[:28]
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        line.append("%s +%.1f gap %.1f dur %.1f" % (short, (s - t0) / 1e3, gap, (e - s) / 1e3))
        prev_end = e
    print(" | ".join(line), "| period %.1f" % ((rows[b][0] - t0) / 1e3))
PY
