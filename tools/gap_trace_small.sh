#!/bin/bash
# tools/gap_trace_small.sh: kernels and the gaps between them for the E-step loop of configs[0]
# (tools/c1_latency.py) -- where the 60-odd microseconds of a small E-step go.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/c1_latency.py
rm -rf /tmp/pgs
rocprofv3 --kernel-trace --output-format csv -d /tmp/pgs -- python3 $R/tools/c1_latency.py > /dev/null 2> /tmp/pgs.err
python3 - $(find /tmp/pgs -name "*kernel_trace.csv" | head -1) > $O/${1:-gap}_small.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:44], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows][-400:]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
    gaps[(n0, n1)].append(s1 - e0)
for n, s, e in seq:
    durs[n].append(e - s)
for k, v in durs.items():
    print("kernel %-46s n=%4d  mean %8.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gaps.items():
    v2 = sorted(v)
    print("gap %-44s -> %-44s n=%4d median %7.1f us" % (k[0], k[1], len(v), v2[len(v2) // 2] / 1e3))
PY
cat $O/${1:-gap}_small.txt
