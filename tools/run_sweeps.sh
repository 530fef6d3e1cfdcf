#!/bin/bash
# tools/run_sweeps.sh SEED [MULT]: all random parity sweeps of tests/sweeps with new seeds (MULT times the
# usual number of cases), summaries to gpurun_out/
S=${1:-101}
M=${2:-1}
O=gpurun_out/sweeps_$S.txt
: > $O
for spec in "stress_small.py $S $((120*M))" "stress_hidden.py $((S+1)) $((120*M))" "stress_em.py $((S+2)) $((40*M))" "stress_gibbs.py $((S+3)) $((30*M))" "stress_reuse.py $((S+4)) $((160*M))" "stress_carry.py $((S+5)) $((60*M))" "stress_many_states.py $((S+6)) $((60*M))" "stress_wide_paths.py $((S+7)) $((40*M))" "many_short.py"; do
  echo "=== $spec" >> $O
  timeout 3000 python3 tests/sweeps/$spec 2>&1 | tail -8 >> $O
  echo "rc=$?" >> $O
done
cat $O
