#!/bin/bash
for i in 1 2; do
for v in "" "BHMM_X_NOEV=1" "BHMM_X_SPIN=1" "BHMM_X_NOEV=1 BHMM_X_SPIN=1"; do
  echo "== $v"
  env $v python3 bench.py --no-cpu --no-secondary --no-steady 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('achieved'))"
done; done
