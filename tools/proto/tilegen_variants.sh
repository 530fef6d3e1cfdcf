#!/bin/bash
# experiment builds of the library with extra macro definitions for the 65..128-state tile kernels (tile_gen.hip +
# tile_gen_nt.hip, one unit per column-tile count):
#   tools/proto/tilegen_variants.sh NAME1 "-DX=1" NAME2 "-DX=2 -DY" ...  -> build_variants/libtg_<NAME>.so  (BHMM_AMD_LIB=...)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
set -e
cd $R/bhmm_amd/csrc
mkdir -p $R/build_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
names=()
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  names+=($n)
  $HIPCC $F $d -c -o $R/build_variants/tile_gen_$n.o tile_gen.hip &
  for nt in 5 6 7 8; do
    $HIPCC $F $d -DTILE_GEN_NT_VALUE=$nt -c -o $R/build_variants/tile_gen_${n}_$nt.o tile_gen_nt.hip &
  done
  wait
done
for n in "${names[@]}"; do
  objs="../lib/obj/bhmm_amd.o ../lib/obj/path_api.o ../lib/obj/wide_api.o ../lib/obj/synth_api.o ../lib/obj/gen_api.o ../lib/obj/big_api.o ../lib/obj/host_model.o ../lib/obj/host_api.o ../lib/obj/comm_api.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libtg_$n.so $objs $R/build_variants/tile_gen_$n.o $R/build_variants/tile_gen_${n}_[5678].o -ldl
done
ls -la $R/build_variants/libtg_*.so
