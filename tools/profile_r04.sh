#!/bin/bash
# tools/profile_r04.sh TAG: round-4 evidence for profiles/r04 (run on the GPU box through gpurun).
#   TAG_bench_5_20.json, TAG_kernel_stats_5_20.csv   the bench line with the driver's flags + rocprofv3
#                                                    --kernel-trace --stats of the same command
#   TAG_c4_time.txt, TAG_c4_kernel_stats.csv         configs[3] (64 states): tile kernels vs the
#                                                    one-segment-per-wavefront kernels, kernel averages
#   TAG_c4_clock.txt                                 shader clock, SQ busy, VALU / MFMA instruction counts
#   TAG_c4_traffic.txt                               FETCH_SIZE / WRITE_SIZE of the tile kernels (separate passes)
#   TAG_gen_time.txt                                 64 and more states: E-step, Viterbi, Gibbs path step
#   TAG_wide_viterbi.txt, TAG_wide_sample.txt        9..64 states over time segments vs the serial kernels,
#   TAG_wide_paths_kernel_stats.csv                  per-kernel averages of both tools
#   TAG_spec_tol.txt                                 headline shape against the boundary tolerance
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
bash $R/tools/profile_driver_flags.sh $tag > $O/${tag}_driver_flags.txt 2>&1
python3 $R/tools/c4_tile.py 1 2 > $O/${tag}_c4_time.txt 2>&1
rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python3 $R/tools/c4_tile.py 1 > /tmp/prof_c4.log 2>&1
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $O/${tag}_c4_kernel_stats.csv
bash $R/tools/pmc_clock.sh tools/c4_tile.py ${tag}_c4 > /dev/null 2>&1
rm -rf /tmp/pm
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/c4_tile.py 1 > /tmp/pm.log 2>&1
python3 - $(find /tmp/pm -name "*counter_collection.csv" | head -1) > $O/${tag}_c4_mfma.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'bhmm::k_tile' in r['Kernel_Name'] or 'bhmm::k_wide_' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k, {c: sum(v[-3:]) / len(v[-3:]) for c, v in cs.items()})
PY
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pt_$ctr -- python3 $R/tools/c4_tile.py 1 > /tmp/pt_$ctr.log 2>&1
done
python3 - $(find /tmp/pt_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pt_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/${tag}_c4_traffic.txt <<'PY'
import csv, sys, collections
def load(p, c):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] == c and ('bhmm::k_tile' in r['Kernel_Name'] or 'bhmm::k_wide_' in r['Kernel_Name']):
            d[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in d.items()}
f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
print("kernel | FETCH_SIZE KB | WRITE_SIZE KB | bytes = 1024 (2 FETCH + WRITE)  (gfx950: FETCH_SIZE counts half of a wide streaming read)")
for k in sorted(set(f) | set(w)):
    print(k, "|", f.get(k), "|", w.get(k), "| %.4g" % (1024.0 * (2 * f.get(k, 0) + w.get(k, 0))))
PY
python3 $R/tools/gen_time.py > $O/${tag}_gen_time.txt 2>&1
python3 $R/tools/wide_viterbi.py 2 > $O/${tag}_wide_viterbi.txt 2>&1
python3 $R/tools/wide_sample.py 4 > $O/${tag}_wide_sample.txt 2>&1
python3 $R/tools/spec_tol.py > $O/${tag}_spec_tol.txt 2>&1
rm -rf /tmp/prof_wp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wp -- python3 $R/tools/wide_paths_once.py > /tmp/prof_wp.log 2>&1
cp $(find /tmp/prof_wp -name "*kernel_stats.csv" | head -1) $O/${tag}_wide_paths_kernel_stats.csv
cat $O/${tag}_driver_flags.txt | tail -6; cat $O/${tag}_c4_time.txt | tail -4; head -8 $O/${tag}_c4_clock.txt; cat $O/${tag}_c4_mfma.txt; cat $O/${tag}_c4_traffic.txt; tail -7 $O/${tag}_gen_time.txt
