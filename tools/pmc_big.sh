#!/bin/bash
# tools/pmc_big.sh N [TAG]: counters of the big kernels (tools/big_time.py N): matrix-pipe busy, waits, texture
# addresser / L2 statistics.  Output: gpurun_out/TAG_big_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-256}
tag=${2:-r05}
out=$R/gpurun_out/${tag}_big_pmc.txt
: > $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pb$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pb$i -- python3 $R/tools/big_time.py $n > /tmp/pb$i.log 2>&1
  python3 - $(find /tmp/pb$i -name "*counter_collection.csv" | head -1) $(find /tmp/pb$i -name "*kernel_trace.csv" | head -1) >> $out <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dd = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if ('k_big_' in k or 'k_tile_' in k) and 'pack' not in k and 'finalize' not in k and 'logl' not in k:
        agg[k[:48]][r["Counter_Name"]].append(float(r['Counter_Value']))
        dd[k[:48]].append(dur.get(r['Dispatch_Id'], 0))
for k, cs in agg.items():
    print(k, "| avg us %.1f |" % (sum(dd[k][-12:]) / max(len(dd[k][-12:]), 1) / 1e3), {c: "%.4g" % (sum(v[-3:]) / len(v[-3:])) for c, v in cs.items()})
PY
done
cat $out
