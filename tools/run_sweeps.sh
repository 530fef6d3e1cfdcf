#!/bin/bash
# tools/run_sweeps.sh SEED: all random parity sweeps of tests/sweeps with new seeds, summaries to gpurun_out/
S=${1:-101}
O=gpurun_out/sweeps_$S.txt
: > $O
for spec in "stress_small.py $S 120" "stress_hidden.py $((S+1)) 120" "stress_em.py $((S+2)) 40" "stress_gibbs.py $((S+3)) 30" "stress_reuse.py $((S+4)) 160" "stress_carry.py $((S+5)) 60" "many_short.py"; do
  echo "=== $spec" >> $O
  timeout 900 python3 tests/sweeps/$spec 2>&1 | tail -4 >> $O
  echo "rc=$?" >> $O
done
cat $O
