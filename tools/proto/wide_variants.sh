#!/bin/bash
# experiment builds of the library with extra macro definitions for wide_api.hip (the 33..64-state tile kernels):
#   tools/proto/wide_variants.sh NAME1 "-DX=1" NAME2 "-DX=2 -DY" ...  -> build_variants/libwide_<NAME>.so  (BHMM_AMD_LIB=...)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
set -e
cd $R/bhmm_amd/csrc
mkdir -p $R/build_variants
names=()
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  names+=($n)
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $d -c -o $R/build_variants/wide_api_$n.o wide_api.hip &
done
wait
for n in "${names[@]}"; do
  objs="../lib/obj/bhmm_amd.o ../lib/obj/path_api.o ../lib/obj/synth_api.o ../lib/obj/gen_api.o ../lib/obj/tile_gen.o ../lib/obj/tile_gen_5.o ../lib/obj/tile_gen_6.o ../lib/obj/tile_gen_7.o ../lib/obj/tile_gen_8.o ../lib/obj/big_api.o ../lib/obj/host_model.o ../lib/obj/host_api.o ../lib/obj/comm_api.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libwide_$n.so $objs $R/build_variants/wide_api_$n.o -ldl
done
ls -la $R/build_variants/libwide_*.so
