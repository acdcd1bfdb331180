#!/bin/bash
# instruction counts + resource usage of the wavefront kernels (CPU-side, no GPU needed)
cd "$(dirname "$0")/../gltf_renderer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include --cuda-device-only -S ${1:-pt_wavefront.hip} -o /tmp/ks_dev.s 2>/dev/null
python3 - <<'PY'
import re
cur=None; counts={}
for l in open('/tmp/ks_dev.s'):
    m=re.match(r'^(_Z\w+):', l)
    if m: cur=m.group(1); counts[cur]=0
    elif cur and l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'): counts[cur]+=1
    if l.startswith('.Lfunc_end'): cur=None
for f,c in counts.items(): print("%-60s %6d instr" % (f[:60], c))
PY
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name):" /tmp/ks_dev.s | paste - - - - | awk '{print $2, "scratch", $4, "sgpr", $6, "vgpr", $8}' | head -20
