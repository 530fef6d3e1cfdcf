#!/bin/bash
# usage: tools/bench_variants.sh name[:chunk] ...   runs bench.py (configs[1]) per library variant
for vc in "$@"; do
  v=${vc%%:*}; c=0; [[ "$vc" == *:* ]] && c=${vc##*:}
  BHMM_AMD_LIB=$PWD/bhmm_amd/lib/variants/libbhmm_amd_$v.so python bench.py --steps 30 --warmup 3 --no-cpu --chunk $c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$vc', d['value'], d['ms_per_step'], d['kernel_ms']['fwdbwd'], d['config']['chunks'])"
done
