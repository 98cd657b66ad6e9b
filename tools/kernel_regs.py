#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels inside a HIP object or shared library (from the code objects' metadata notes).

    python tools/kernel_regs.py avt_amd/csrc/gemm_persist.o [name-substring]
"""
import os, re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'


def kernel_table(path):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run([f'{LLVM}/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
        data = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        pos = [m.start() for m in re.finditer(re.escape(magic), data)]
        for i, p in enumerate(pos):
            b, o = os.path.join(td, f'b{i}.bin'), os.path.join(td, f'co{i}.o')
            open(b, 'wb').write(data[p:pos[i + 1] if i + 1 < len(pos) else len(data)])
            subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={b}',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={o}'], check=True, capture_output=True)
            txt = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', o], check=True, capture_output=True, text=True).stdout
            cur = {}
            for ln in txt.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', ln)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == 'agpr_count' and cur.get('name'):
                    rows.append(cur); cur = {}
                if k in ('agpr_count', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size',
                         'private_segment_fixed_size', 'name'):
                    cur[k] = v
            if cur.get('name'):
                rows.append(cur)
    return rows


if __name__ == '__main__':
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for r in kernel_table(sys.argv[1]):
        name = subprocess.run(['c++filt', r.get('name', '?')], capture_output=True, text=True).stdout.strip()
        if sub in name:
            print(f"{name[:90]:90s} vgpr {r.get('vgpr_count')} agpr {r.get('agpr_count')} sgpr {r.get('sgpr_count')} "
                  f"spill v{r.get('vgpr_spill_count')} s{r.get('sgpr_spill_count')} scratch {r.get('private_segment_fixed_size')} lds {r.get('group_segment_fixed_size')}")
