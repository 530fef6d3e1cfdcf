#!/bin/bash
# tools/profile_c4.sh TAG: evidence for the wide family on BASELINE configs[3] (64-state Gaussian,
# 128 x 1e5): rocprofv3 kernel statistics and two SQ counter passes of tools/c4_once.py.
# Writes gpurun_out/TAG_c4_*.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/c4_once.py > $O/${tag}_c4_time.txt 2> /tmp/c4.err < /dev/null || tail -3 /tmp/c4.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4k -- python3 $R/tools/c4_once.py > /tmp/c4k.log 2>&1 < /dev/null
f=$(find /tmp/c4k -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/${tag}_c4_kernel_stats.csv
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/c4p$i -- python3 $R/tools/c4_once.py > /tmp/c4p$i.log 2>&1 < /dev/null
  f=$(find /tmp/c4p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" > $O/${tag}_c4_pmc_set$i.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    kn = r['Kernel_Name']
    if 'k_wide_fwd' in kn or 'k_wide_bwd' in kn:
        tag = ('fwd' if 'k_wide_fwd' in kn else 'bwd') + ('(lazy,64)' if 'true, true' in kn else '(serial plan)')
        d[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
for (t, k), v in sorted(d.items()):
    print(t, k, len(v), sum(v) / len(v))
PY
  else tail -5 /tmp/c4p$i.log > $O/${tag}_c4_pmc_set$i.err; fi
done
