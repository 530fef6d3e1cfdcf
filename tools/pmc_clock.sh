#!/bin/bash
# tools/pmc_clock.sh SCRIPT.py [TAG]: effective shader clock of every kernel of a python script =
# GRBM_GUI_ACTIVE / duration (MI355X_MICROARCH.md "DVFS give-back"), with the SQ busy / wave cycle
# and VALU instruction counters of the same pass.  Output: gpurun_out/TAG_clock.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${2:-clock}
rm -rf /tmp/pc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pc -- python3 $R/$1 > /tmp/pc.log 2>&1
c=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
k=$(find /tmp/pc -name "*kernel_trace.csv" | head -1)
python3 - "$c" "$k" > $R/gpurun_out/${tag}_clock.txt <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    ctr[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.defaultdict(list)
for d, cs in ctr.items():
    if d in dur:
        ns, name = dur[d]
        agg[name[:60]].append((ns, cs))
print("kernel | calls | avg us | eff. clock GHz (GRBM_GUI_ACTIVE/ns) | SQ_BUSY_CYCLES | SQ_WAVE_CYCLES | SQ_WAVES | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU")
for name, rows in sorted(agg.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
    rows = rows[-3:] if len(rows) > 3 else rows
    ns = sum(r[0] for r in rows) / len(rows)
    def avg(c): return sum(r[1].get(c, 0.0) for r in rows) / len(rows)
    print("%s | %d | %.1f | %.3f | %.4g | %.4g | %.4g | %.4g | %.4g" % (
        name, len(rows), ns / 1e3, avg('GRBM_GUI_ACTIVE') / ns, avg('SQ_BUSY_CYCLES'),
        avg('SQ_WAVE_CYCLES'), avg('SQ_WAVES'), avg('SQ_INSTS_VALU'), avg('SQ_ACTIVE_INST_VALU')))
PY
tail -3 /tmp/pc.log
cat $R/gpurun_out/${tag}_clock.txt | head -12
