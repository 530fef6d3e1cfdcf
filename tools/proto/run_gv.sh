cd /tmp; export TMPDIR=/tmp
for v in noloop nosum; do
  rm -rf /tmp/gv_$v
  BHMM_AMD_LIB=$GRAFT_REPO_ROOT/build_variants/libgen_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gv_$v -- python3 $GRAFT_REPO_ROOT/tools/vit_once.py 192 > /dev/null 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/proto/kern_avgs.py /tmp/gv_$v viterbi_rows
done
