#!/bin/bash
# tools/proto/run_wv.sh NAME...: rocprofv3 kernel averages of tools/c4_once.py under build_variants/libwide_<NAME>.so
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/wv_$v
  BHMM_AMD_LIB=$GRAFT_REPO_ROOT/build_variants/libwide_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wv_$v -- python3 $GRAFT_REPO_ROOT/tools/c4_once.py > /tmp/wv_$v.log 2>&1
  echo "== $v: $(grep '^ms' /tmp/wv_$v.log)"; python3 $GRAFT_REPO_ROOT/tools/proto/kern_avgs.py /tmp/wv_$v k_tile_fwd k_tile_bwd
done
