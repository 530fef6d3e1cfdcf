#!/bin/bash
# builds libbhmm_amd.so variants with WVS_LDS_ROWS = 0, 1, 3, 4 into build_variants/ldsN/ (the default build has 3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/bhmm_amd/csrc
for r in 2 4; do
  mkdir -p $R/build_variants/lds$r
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -ffp-contract=off -DWVS_LDS_ROWS=$r -c -o $R/build_variants/lds$r/path_api.o path_api.hip &
done
wait
for r in 2 4; do
  objs=$(ls ../lib/obj/*.o | grep -v path_api.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lds$r/libbhmm_amd.so $objs $R/build_variants/lds$r/path_api.o -ldl
done
ls -la $R/build_variants/lds*/libbhmm_amd.so
