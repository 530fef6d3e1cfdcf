#!/bin/bash
# tools/run_sweeps_r06.sh SEED [MULT]: the sweeps that reach the code paths of round 6, in the forms that force them:
#   (a) the margin rule + mending at every state count (BHMM_AMD_VIT_MARGIN_FORCE=1)
#   (b) a wide draw watch: every draw within 1e-4 of a cumulative-sum edge is decided again on the serial recursion
#   (c) ... and treated as a decision that did not stand: the call repeated on exact alpha rows
# summaries to gpurun_out/sweeps_r06_SEED.txt
S=${1:-6201}
M=${2:-1}
O=gpurun_out/sweeps_r06_$S.txt
: > $O
run() { echo "=== $1 | $2" >> $O; env $1 timeout 3000 python3 tests/sweeps/$2 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|B /= B" | tail -4 >> $O; echo "rc=$?" >> $O; }
run "BHMM_AMD_VIT_MARGIN_FORCE=1" "stress_wide_paths.py $S $((60*M))"
run "BHMM_AMD_VIT_MARGIN_FORCE=1" "stress_many_states.py $((S+1)) $((60*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-4" "stress_wide_paths.py $((S+2)) $((60*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-4" "stress_small.py $((S+3)) $((120*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-4" "stress_gibbs.py $((S+4)) $((30*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-4" "stress_many_states.py $((S+5)) $((60*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-3 BHMM_AMD_DRAW_TEST_REDO=1" "stress_wide_paths.py $((S+6)) $((40*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-3 BHMM_AMD_DRAW_TEST_REDO=1" "stress_small.py $((S+7)) $((80*M))"
run "BHMM_AMD_DRAW_WATCH_TOL=1e-3 BHMM_AMD_DRAW_TEST_REDO=1" "stress_gibbs.py $((S+8)) $((20*M))"
cat $O
