"""tools/traffic_r05.py FETCH.csv WRITE.csv CLOCK.txt TAG -> the JSON bench.py quotes as roofline.traffic
(profiles/traffic_current.json): per workload the HBM bytes of the two sweep launches of one E-step
and their SQ_INSTS_VALU, from the rocprofv3 counter passes of tools/profile_r05.sh."""
import collections
import csv
import json
import sys


def load(path, ctr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == ctr:
            k = r['Kernel_Name']
            if 'bhmm::' in k or 'opyBuffer' in k:
                d[k[:110]].append(float(r['Counter_Value']))
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in d.items()}, {k: len(v) for k, v in d.items()}


f, nf = load(sys.argv[1], 'FETCH_SIZE')
w, nw = load(sys.argv[2], 'WRITE_SIZE')
tag = sys.argv[4]
valu = {}
for ln in open(sys.argv[3]):
    p = [x.strip() for x in ln.split('|')]
    if len(p) >= 9 and p[0] != 'kernel':
        clk = float(p[3])
        valu[p[0]] = (float(p[7]), float(p[2]), clk / 8.0 if clk > 4.0 else clk)   # SQ_INSTS_VALU, avg us, clock GHz
        # (GRBM_GUI_ACTIVE comes summed over the 8 XCDs from this rocprofv3)
out = {"unit": "FETCH_SIZE / WRITE_SIZE in KB per launch (mean of the last 3 launches); bytes = 1024 * (2 * FETCH_SIZE + "
               "WRITE_SIZE): on gfx950 FETCH_SIZE reports half of the bytes of a wide streaming read (MI355X_MICROARCH.md, "
               "HBM) -- checked on the 1 GiB clone below (expected FETCH 524288 KB = half, WRITE 1048576 KB)",
       "source": "profiles/%s/%s_traffic.json (tools/profile_r05.sh %s: tools/pmc_r05.py under rocprofv3 --pmc, one "
                 "counter per pass; SQ_INSTS_VALU from profiles/%s/%s_clock.txt)" % (tag[:3], tag, tag, tag[:3], tag),
       "issue_source": "one vector instruction per SIMD and four cycles (fp64 FMA, DPP move and integer instructions alike: "
                       "tools/ubench/issue_rate.hip, profiles/r02/r02_ubench_issue_rate.txt) at the shader clock of the same "
                       "counter pass, GRBM_GUI_ACTIVE / duration / 8 XCDs",
       "workloads": {}, "kernels": {}}
for k in sorted(set(f) | set(w)):
    out["kernels"][k] = {"FETCH_SIZE_KB": f.get(k), "WRITE_SIZE_KB": w.get(k), "launches_seen": nf.get(k, nw.get(k)),
                         "bytes": 1024.0 * (2.0 * f.get(k, 0.0) + w.get(k, 0.0))}
# template arguments: <N, KIND (0 gaussian, 1 discrete), SPEC, ..., PHASE (2 = P1 of k_estep_light, 3 = P2)>
for key, kind, shape, balg in (("configs2", 1, (1024, 1000000), 136), ("configs1", 0, (256, 100000), 144)):
    p1 = [(k, v) for k, v in out["kernels"].items() if 'k_estep_light<8, %d, true' % kind in k and ', 2>' in k]
    p2 = [(k, v) for k, v in out["kernels"].items() if 'k_estep<8, %d, true' % kind in k and ', 3>' in k]
    if not (p1 and p2):
        continue
    ent = {"shape": list(shape), "kernels": [p1[0][0], p2[0][0]],
           "traffic_bytes_per_launch": p1[0][1]["bytes"] + p2[0][1]["bytes"],
           "algorithmic_bytes_per_launch": balg * shape[0] * shape[1]}
    v1 = [v for k, v in valu.items() if p1[0][0].startswith(k[:58]) or k.startswith(p1[0][0][:58])]
    v2 = [v for k, v in valu.items() if p2[0][0].startswith(k[:58]) or k.startswith(p2[0][0][:58])]
    if v1 and v2:
        i1, i2 = v1[0][0], v2[0][0]
        ent["valu_wave_insts_per_launch"] = i1 + i2
        ent["valu_wave_insts"] = {"P1": i1, "P2": i2}
        # issue ceiling of a SIMD: one vector instruction per four cycles at the clock the counters saw
        # (GRBM_GUI_ACTIVE / duration), weighted by the two kernels' instruction counts
        ent["issue_ceiling_insts_per_us_per_simd"] = (i1 + i2) / (i1 / (250.0 * v1[0][2]) + i2 / (250.0 * v2[0][2]))
        ent["under_counters"] = {"P1_us": v1[0][1], "P2_us": v2[0][1], "P1_clock_GHz": v1[0][2], "P2_clock_GHz": v2[0][2]}
    out["workloads"][key] = ent
print(json.dumps(out, indent=1))
