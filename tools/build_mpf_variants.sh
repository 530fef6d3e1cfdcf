#!/bin/bash
# builds libbhmm_amd.so variants with SMP_MPF_VALUE = 2, 4 (k_smp_maps: load-ahead of the fp32 alpha rows) into build_variants/mpfN/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd $R/bhmm_amd/csrc
for r in 2 4; do
  mkdir -p $R/build_variants/mpf$r
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -ffp-contract=off -DSMP_MPF_VALUE=$r -c -o $R/build_variants/mpf$r/path_api.o path_api.hip &
done
wait
for r in 2 4; do
  objs=$(ls ../lib/obj/*.o | grep -v path_api.o)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/mpf$r/libbhmm_amd.so $objs $R/build_variants/mpf$r/path_api.o -ldl
done
ls -la $R/build_variants/mpf*/libbhmm_amd.so
