#!/bin/bash
# tools/profile_round2.sh TAG: the per-round evidence for profiles/ (run on the GPU box through
# gpurun).  Writes gpurun_out/TAG_*:
#   TAG_bench.json               the bench line (N = 1, CPU legs and secondary configs included)
#   TAG_kernel_stats.csv         rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu`
#   TAG_bench_under_rocprof.json the line that run printed
#   TAG_traffic.json             HBM bytes per launch of every bhmm kernel of E-step, Gibbs sweep and
#                                Viterbi at the configs[1] shape: separate FETCH_SIZE / WRITE_SIZE
#                                passes, FETCH doubled (gfx950), calibrated on a 1 GiB copy
#   TAG_clock.txt                effective shader clock + SQ busy / wave cycles / VALU instructions
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python3 $R/bench.py --no-cpu > $O/${tag}_bench_under_rocprof.json 2> /tmp/prof_k.err
cp $(find /tmp/prof_k -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
# the headline loop alone (50 + 200 launches of the sweep pair, nothing else): the averages that
# roofline.achieved of the bench line must agree with
rm -rf /tmp/prof_h
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -- python3 $R/bench.py --no-cpu --no-secondary --no-steady > $O/${tag}_bench_headline_under_rocprof.json 2> /tmp/prof_h.err
cp $(find /tmp/prof_h -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats_headline.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$ctr -- python3 $R/tools/pmc_traffic2.py > /tmp/prof_$ctr.log 2>&1
done
python3 - $(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/${tag}_traffic.json <<'PY'
import csv, sys, json, collections
def load(path, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == ctr:
            k = r['Kernel_Name']
            if 'bhmm::' in k or 'opyBuffer' in k or 'elementwise' in k:
                d[k[:110]].append(float(r['Counter_Value']))
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in d.items()}, {k: len(v) for k, v in d.items()}
f, nf = load(sys.argv[1], 'FETCH_SIZE')
w, nw = load(sys.argv[2], 'WRITE_SIZE')
out = {"unit": "FETCH_SIZE / WRITE_SIZE in KB per launch (mean of the last 3 launches); bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): "
               "on gfx950 FETCH_SIZE reports half of the bytes of a wide streaming read (MI355X_MICROARCH.md, HBM) -- checked on the "
               "1 GiB clone below (expected FETCH 524288 KB = half, WRITE 1048576 KB)",
       "workload": "tools/pmc_traffic2.py: configs[1] shape (8-state Gaussian, 256 x 100000): 3 E-steps, 3 Gibbs hidden-path sweeps, 3 Viterbi passes, 3 x 1 GiB clone",
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    out["kernels"][k] = {"FETCH_SIZE_KB": f.get(k), "WRITE_SIZE_KB": w.get(k), "launches_seen": nf.get(k, nw.get(k)),
                         "bytes": 1024.0 * (2.0 * f.get(k, 0.0) + w.get(k, 0.0))}
p1 = [v for k, v in out["kernels"].items() if 'k_estep_light' in k and ', 2>' in k]
p2 = [v for k, v in out["kernels"].items() if 'k_estep<' in k and ', 3>' in k]
if p1 and p2:
    out["traffic_bytes_per_launch"] = p1[0]["bytes"] + p2[0]["bytes"]
    out["algorithmic_bytes_per_launch"] = 144 * 256 * 100000
    out["kernel"] = "k_estep_light<8,gauss,spec,PH_P1> + k_estep<8,gauss,spec,PH_P2> (one E-step sweep = these two launches)"
print(json.dumps(out, indent=1))
PY
bash $R/tools/pmc_clock.sh tools/pmc_traffic2.py ${tag} > /dev/null 2>&1
head -c 600 $O/${tag}_bench.json; echo
head -30 $O/${tag}_clock.txt
