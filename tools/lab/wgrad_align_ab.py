"""Whole training step with the weight gradients' automatic split-K factor against XCD-aligned factors (judge's round-5 item 5): an output with at most
32 tiles of 256 x 256 takes 8 * floor(32 / tiles) splits, so that every XCD owns whole splits (qkv: 27 tiles x 8 = 216 workgroups instead of 243,
proj: 9 x 24 = 216 instead of 252) and every operand panel crosses the fabric once.  The factor is an argument of avt_gemm_accum_bf16: the A/B patches
the ctypes call, not the library.   usage: python tools/lab/wgrad_align_ab.py {auto|align} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import lib
mode = sys.argv[1]
if mode == 'align':
    real = lib.call
    def call(name, *a):
        if name == 'avt_gemm_accum_bf16' and a[9] == 0:
            M, N, K = a[6], a[7], a[8]
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            if tiles <= 32 and K >= 64 * 8 * 8 * (32 // tiles):
                a = a[:9] + (8 * (32 // tiles),) + a[10:]
        return real(name, *a)
    lib.call = call
import bench
bench.main(sys.argv[2:])
