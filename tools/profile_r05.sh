#!/bin/bash
# tools/profile_r05.sh TAG: round-5 evidence for profiles/r05 (run on the GPU box through gpurun).
#   TAG_bench_5_20.json                 the bench line with the driver's flags (--gpus 1 --steps 20 --warmup 5)
#   TAG_c2_kernel_stats.csv             rocprofv3 --kernel-trace --stats of the same command with
#   TAG_bench_5_20_under_rocprof.json   --no-cpu --no-secondary --no-steady: the configs[2] sweep launches
#                                       of exactly the calibration + warm-up + timed steps
#   TAG_traffic.json                    HBM bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes, FETCH
#                                       doubled per MI355X_MICROARCH.md, checked on a 1 GiB copy) and
#                                       SQ_INSTS_VALU of the sweep launches of configs[2] AND configs[1]
#                                       (tools/pmc_r05.py) -- the file bench.py quotes
#   TAG_clock.txt                       shader clock / SQ counters of the same kernels
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_5_20.json 2> /tmp/b.err || tail -5 /tmp/b.err
rm -rf /tmp/prof_d
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-secondary --no-steady > $O/${tag}_bench_5_20_under_rocprof.json 2> /tmp/prof_d.err
cp $(find /tmp/prof_d -name "*kernel_stats.csv" | head -1) $O/${tag}_c2_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -- python3 $R/tools/pmc_r05.py > /tmp/prof_$ctr.log 2>&1
  tail -2 /tmp/prof_$ctr.log
done
bash $R/tools/pmc_clock.sh tools/pmc_r05.py ${tag} > /dev/null 2>&1
python3 $R/tools/traffic_r05.py $(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/${tag}_clock.txt $tag > $O/${tag}_traffic.json
python3 - $O/${tag}_bench_5_20.json $O/${tag}_bench_5_20_under_rocprof.json $O/${tag}_c2_kernel_stats.csv <<'PY'
import json, sys, csv
for f in sys.argv[1:3]:
    d = json.load(open(f)); print(f.split('/')[-1], "ms_per_step %.4f" % d["ms_per_step"], "sweep by HIP events %.4f ms" % d["kernel_ms"]["fwdbwd"], "frac %.3f" % d["roofline"]["frac"], d["config"]["workload"])
for r in csv.DictReader(open(sys.argv[3])):
    if "k_estep" in r["Name"]:
        print(r["Name"][:64], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
head -c 1500 $O/${tag}_traffic.json | head -40
head -12 $O/${tag}_clock.txt
