for P in 4 8 16; do
  export BHMM_AMD_SMP_PARTS=$P
  bash tools/ktrace.sh tools/c5_once.py > gpurun_out/r02n_parts$P.txt 2>&1
  echo "P=$P"; grep -E "k_smp|k_estep_light" gpurun_out/r02n_parts$P.txt
done
