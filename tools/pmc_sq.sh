#!/bin/bash
# SQ counter passes over the E-step kernel; output: gpurun_out/pmc_<set>.csv (k_estep rows only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -- python3 $R/tools/pmc_estep.py > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('$f')):
    kn = r['Kernel_Name']
    tag = 'P1' if 'k_estep_light' in kn else ('P2' if 'k_estep' in kn else None)
    if tag: d[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
for (t,k),v in sorted(d.items()): print(t, k, len(v), sum(v)/len(v))
" > $R/gpurun_out/pmc_set$i.txt; else tail -5 /tmp/pmc$i.log > $R/gpurun_out/pmc_set$i.err; fi
done
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ[C]*_[A-Z_0-9]*" | sort -u > $R/gpurun_out/sq_counters.txt
