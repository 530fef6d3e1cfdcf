"""rocprofv3 kernel averages of one command's csv:  python tools/proto/kern_avgs.py <dir> [substr ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    nm = r["Name"].split("(")[0]
    if not sys.argv[2:] or any(k in nm for k in sys.argv[2:]):
        print("   %-64s %5s %9.1f us" % (nm[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
