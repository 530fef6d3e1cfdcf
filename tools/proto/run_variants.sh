#!/bin/bash
# tools/proto/run_variants.sh PREFIX "n1 n2" NAME...: rocprofv3 kernel averages of tools/big_time.py under each build_variants/lib<PREFIX>_<NAME>.so
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; pre=$1; ns=$2; shift 2
for v in "$@"; do
  export BHMM_AMD_LIB=$R/build_variants/lib${pre}_$v.so
  rm -rf /tmp/pv_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$v -- python3 $R/tools/big_time.py $ns > /tmp/pv_$v.log 2>&1
  echo "== $v"; grep "^n=" /tmp/pv_$v.log | cut -c1-150
  python3 - $(find /tmp/pv_$v -name "*kernel_stats.csv") <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    nm=r['Name'].split('(')[0]
    if 'k_tile_bwd' in nm or 'k_tile_fwd' in nm or 'xi_gemm' in nm:
        print("   %-60s %4s %9.1f us"%(nm[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
