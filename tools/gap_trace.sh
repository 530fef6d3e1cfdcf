#!/bin/bash
# tools/gap_trace.sh: where the time between the kernels of one E-step goes -- start / end timestamps of
# every dispatch of `bench.py --no-cpu --no-secondary --no-steady` (rocprofv3 --kernel-trace), gaps between
# consecutive kernels of the timed loop.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pgap
rocprofv3 --kernel-trace --output-format csv -d /tmp/pgap -- python3 $R/bench.py --no-cpu --no-secondary --no-steady > /dev/null 2> /tmp/pgap.err
python3 - $(find /tmp/pgap -name "*kernel_trace.csv" | head -1) > $O/${1:-gap}_trace.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# the steady part: last 600 dispatches
seq = seq[-600:]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
    gaps[(n0, n1)].append(s1 - e0)
for n, s, e in seq:
    durs[n].append(e - s)
for k, v in durs.items():
    print("kernel %-42s n=%4d  mean %8.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gaps.items():
    v2 = sorted(v)
    print("gap %-40s -> %-40s n=%4d median %7.1f us  mean %7.1f us" % (k[0], k[1], len(v), v2[len(v2) // 2] / 1e3, sum(v) / len(v) / 1e3))
PY
cat $O/${1:-gap}_trace.txt
