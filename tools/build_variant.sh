#!/bin/bash
# tools/build_variant.sh NAME [-Dflags...]: builds bhmm_amd/lib/variants/libbhmm_amd_NAME.so with
# extra macro definitions for the E-step translation unit (kernel experiments; select at run
# time with BHMM_AMD_LIB=<path>).
set -e
cd "$(dirname "$0")/../bhmm_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -c -o /tmp/bhmm_var_$name.o bhmm_amd.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libbhmm_amd_$name.so /tmp/bhmm_var_$name.o ../lib/obj/path_api.o ../lib/obj/wide_api.o ../lib/obj/synth_api.o ../lib/obj/host_mstep.o
echo built $name
