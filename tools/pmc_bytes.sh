#!/bin/bash
# tools/pmc_bytes.sh SCRIPT.py TAG: HBM bytes per launch of the bhmm kernels of a python script from
# separate FETCH_SIZE / WRITE_SIZE passes (FETCH doubled: gfx950, MI355X_MICROARCH.md section HBM).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pb_$ctr -- python3 $R/$1 > /tmp/pb_$ctr.log 2>&1
done
python3 - $(find /tmp/pb_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pb_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $R/gpurun_out/$2_bytes.txt <<'PY'
import csv, sys, collections
def load(path, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == ctr and 'bhmm::' in r['Kernel_Name']:
            d[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in d.items()}
f = load(sys.argv[1], 'FETCH_SIZE'); w = load(sys.argv[2], 'WRITE_SIZE')
print("kernel | FETCH_SIZE KB | WRITE_SIZE KB | bytes = 1024 (2 FETCH + WRITE) in GB")
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
    print("%s | %.0f | %.0f | %.3f" % (k, f.get(k, 0), w.get(k, 0), 1024 * (2 * f.get(k, 0) + w.get(k, 0)) / 1e9))
PY
head -8 $R/gpurun_out/$2_bytes.txt
