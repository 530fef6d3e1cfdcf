#!/bin/bash
# experiment builds of the library: what does a piece of the tile step cost?  (tools/proto, not shipped)
#   tools/proto/variants.sh NOEMIT NOSTORE NOSTATS NOXI ...  -> build_variants/lib_<NAME>.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
set -e
cd $R/bhmm_amd/csrc
mkdir -p $R/build_variants
for v in "$@"; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DTILE_X_$v -c -o $R/build_variants/wide_api_$v.o wide_api.hip &
done
wait
for v in "$@"; do
  objs="../lib/obj/bhmm_amd.o ../lib/obj/path_api.o ../lib/obj/synth_api.o ../lib/obj/gen_api.o ../lib/obj/tile_gen.o ../lib/obj/host_model.o ../lib/obj/host_api.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/lib_$v.so $objs $R/build_variants/wide_api_$v.o
done
ls -la $R/build_variants/*.so
