#!/bin/bash
# compile wide_api.hip to ISA and list the register use of the tile kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd $R/bhmm_amd/csrc && $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -S --cuda-device-only -o /tmp/wide_api.s wide_api.hip 2>&1 | grep -E "error" -A5 | head -40
python3 - <<'PY'
import re
t=open('/tmp/wide_api.s').read()
for m in re.finditer(r"\.name:\s+(\S*k_tile_\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", t):
    print(m.group(1)[:60], "scratch", m.group(2), "vgpr", m.group(3), "spills", m.group(4))
PY
for k in fwdILi4ELi0 bwdILi4ELi0; do awk "/^_ZN4bhmm10k_tile_$k/{f=1} f{print} /s_endpgm/{if(f) exit}" /tmp/wide_api.s > /tmp/tile_$k.s; awk '/Loop Header/{name=$1; start[name]=NR} /scratch_/{c[name]++} /v_mfma/{m[name]++} END{for(k in start) if (m[k]>0) print start[k], k, "scratch", c[k]+0, "mfma", m[k]+0}' /tmp/tile_$k.s | sort -n | tr '\n' ';'; echo; done
