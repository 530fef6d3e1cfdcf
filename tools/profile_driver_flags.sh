#!/bin/bash
# tools/profile_driver_flags.sh TAG: rocprofv3 --kernel-trace --stats of the bench line with the
# flags the round driver uses (--gpus 1 --steps 20 --warmup 5): the first ~30 E-steps after the GPU
# idled run slower, so this line and its kernel averages differ from the
# steady-state ones of profile_round2.sh (50 + 200 steps).
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_5_20.json 2> /tmp/b.err || tail -5 /tmp/b.err
rm -rf /tmp/prof_d
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-secondary --no-steady > $O/${tag}_bench_5_20_under_rocprof.json 2> /tmp/prof_d.err
cp $(find /tmp/prof_d -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats_5_20.csv
python3 - $O/${tag}_bench_5_20.json $O/${tag}_bench_5_20_under_rocprof.json $O/${tag}_kernel_stats_5_20.csv <<'PY'
import json, sys, csv
for f in sys.argv[1:3]:
    d = json.load(open(f)); print(f.split('/')[-1], "ms_per_step %.4f" % d["ms_per_step"], "sweep by HIP events %.4f ms" % d["kernel_ms"]["fwdbwd"], "frac %.3f" % d["roofline"]["frac"])
for r in csv.DictReader(open(sys.argv[3])):
    if "k_estep" in r["Name"]:
        print(r["Name"][:64], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
