#!/bin/bash
# kernel resource usage summary (VGPR / AGPR / SGPR / spills / occupancy) of the E-step TU
cd "$(dirname "$0")/bhmm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c ${1:-bhmm_amd.hip} -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | awk '/Function Name/{name=$NF} /remark:.*VGPRs:/{v=$NF} /AGPRs:/{a=$NF} /TotalSGPRs/{sg=$NF} /ScratchSize/{sc=$NF} /Occupancy/{oc=$NF} /SGPRs Spill/{ss=$NF} /VGPRs Spill/{vs=$NF} /LDS Size/{printf "%-90s vgpr=%s agpr=%s sgpr=%s sspill=%s vspill=%s scratch=%s occ=%s\n", name, v, a, sg, ss, vs, sc, oc}' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | grep -E "${2:-Li8}"
