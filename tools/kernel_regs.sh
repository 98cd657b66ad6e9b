#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> <kernel-name-substring>: VGPR / scratch / spill counts per instantiation
mkdir -p /tmp/asm && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -std=c++17 -I../../include -S --cuda-device-only $1 -o /tmp/asm/$(basename $1 .hip).s 2>&1 | grep -v "warning\|^$" | tail -3
python3 - "$1" "$2" <<'PY'
import re, sys
s = open('/tmp/asm/' + sys.argv[1].split('/')[-1].replace('.hip', '.s')).read()
for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    if sys.argv[2] in blk:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
        print(name[:70], 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'sgpr', g('sgpr_count'), 'scratch', g('private_segment_fixed_size'), 'spill', g('vgpr_spill_count'))
PY
