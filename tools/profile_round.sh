#!/bin/bash
# tools/profile_round.sh TAG: the per-round evidence for profiles/ (run on the GPU box through
# gpurun).  Writes gpurun_out/TAG_*: bench line, rocprofv3 kernel statistics of the same command,
# HBM traffic of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes, calibrated on a
# 1 GiB copy in the same run) and a few SQ counters.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $O/${tag}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/${tag}_bench_under_rocprof.json 2> /tmp/prof_k.err
cp $(find /tmp/prof_k -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -- python3 $R/tools/pmc_traffic.py > /tmp/prof_$ctr.log 2>&1
  f=$(find /tmp/prof_$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr > $O/${tag}_pmc_$ctr.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    name = ('k_estep_light(P1)' if 'k_estep_light' in k else 'k_estep(P2/ALL)' if 'k_estep' in k
            else ('copy' if 'opy' in k else None))
    if name and r['Counter_Name'] == sys.argv[2]:
        d[name].append(float(r['Counter_Value']))
for k, v in d.items():
    print(k, sys.argv[2], len(v), sum(v[-3:]) / len(v[-3:]))
PY
done
bash $R/tools/pmc_sq.sh
cat $O/pmc_set*.txt > $O/${tag}_pmc_SQ.txt
