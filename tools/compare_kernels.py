#!/usr/bin/env python3
"""Kernel-by-kernel comparison of the device code of two builds of the library: which kernels exist in both, and which of those have the same
instruction stream (PC-relative literals of `s_add_u32 / s_addc_u32` -- the addresses of device globals such as the GELU table, which move when
other kernels come or go -- are masked).  Round 6 used it to show that taking the lab hooks / A-B switches out of the sources changed no instruction
of any kernel that stayed (190 of 190; the 11 that left: the two-phase attention backward and the two-workgroups-per-CU GEMM).
    python tools/compare_kernels.py before.so after.so"""
import hashlib
import re
import sys

import isa_async_check as I


def kernels(so):
    out = {}
    for _, lines in I.device_disassembly(so):
        cur = None
        for ln in lines:
            m = re.match(r'^[0-9a-f]+ <(.+)>:$', ln)
            if m:
                cur = m.group(1)
                out[cur] = hashlib.sha1()
                continue
            if cur and '\t' in ln:
                ins = ln.split('//')[0].strip()
                ins = re.sub(r'^(s_addc?_u32 s\d+, s\d+, )0x[0-9a-f]{5,8}$', r'\1<pcrel>', ins)
                out[cur].update((ins + '\n').encode())
    return {k: v.hexdigest() for k, v in out.items()}


if __name__ == '__main__':
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same = [k for k in a if k in b and a[k] == b[k]]
    diff = [k for k in a if k in b and a[k] != b[k]]
    print(f'kernels: {len(a)} before, {len(b)} after; in both {len(same) + len(diff)}: identical instruction streams {len(same)}, different {len(diff)}')
    for k in sorted(set(a) - set(b)):
        print('  only before:', k)
    for k in sorted(set(b) - set(a)):
        print('  only after: ', k)
    for k in diff:
        print('  different:  ', k)
    sys.exit(1 if diff else 0)
