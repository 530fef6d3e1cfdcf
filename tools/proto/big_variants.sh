#!/bin/bash
# experiment builds of the library for the kernels of big_kernels.hpp: what does a piece of the step cost?
#   tools/proto/big_variants.sh NOEMIT NOSTORE NOSTREAM NOLOG ...  -> build_variants/libbig_<NAME>.so  (BHMM_AMD_LIB=...)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
set -e
cd $R/bhmm_amd/csrc
mkdir -p $R/build_variants
for v in "$@"; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DBIG_X_$v -c -o $R/build_variants/big_api_$v.o big_api.hip &
done
wait
for v in "$@"; do
  objs="../lib/obj/bhmm_amd.o ../lib/obj/path_api.o ../lib/obj/wide_api.o ../lib/obj/synth_api.o ../lib/obj/gen_api.o ../lib/obj/tile_gen.o ../lib/obj/tile_gen_5.o ../lib/obj/tile_gen_6.o ../lib/obj/tile_gen_7.o ../lib/obj/tile_gen_8.o ../lib/obj/host_model.o ../lib/obj/host_api.o ../lib/obj/comm_api.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libbig_$v.so $objs $R/build_variants/big_api_$v.o -ldl
done
ls -la $R/build_variants/libbig_*.so
