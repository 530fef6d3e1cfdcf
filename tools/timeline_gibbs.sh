#!/bin/bash
# tools/timeline_gibbs.sh: kernel timeline of the Gibbs hidden-path sweep (tools/gibbs_var.py) --
# start offsets, gaps and durations of the kernels between two forward-only launches (last 6 sweeps).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlg -- python3 $GRAFT_REPO_ROOT/tools/gibbs_var.py > /tmp/tlg.log 2>&1
f=$(find /tmp/tlg -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_estep_light' in r[2] and ', 1>' in r[2]]
idx = idx[-7:]
for a, b in zip(idx[:-1], idx[1:]):
    t0 = rows[a][0]
    line, prev_end = [], None
    for s, e, n in rows[a:b]:
        short = n.split('(')[0].replace('void bhmm::', '').replace('bhmm::', '').replace('__amd_rocclr_', '')[:22]
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        line.append("%s gap %.1f dur %.1f" % (short, gap, (e - s) / 1e3))
        prev_end = e
    print(" | ".join(line), "| period %.1f" % ((rows[b][0] - t0) / 1e3))
PY
