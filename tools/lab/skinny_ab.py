"""Whole training step with the skinny kernel (the library's choice for <= 32 output rows) against the 64 x 64 tiles those GEMMs took before round 6:
the A/B patches the tile argument of the ctypes call.   usage: python tools/lab/skinny_ab.py {skinny|tiles64} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import lib
if sys.argv[1] == 'tiles64':
    real = lib.call
    def call(name, *a):
        if name in ('avt_gemm_bf16', 'avt_gemm_ln_bf16') and a[25] == 0 and a[23] in (0, 1) and a[1] and a[8] <= 64:
            M, N = a[8], a[9]
            a = a[:25] + (643 if (a[4] or ((M + 63) // 64) * ((N + 63) // 64) <= 256) else 64,) + a[26:]
        return real(name, *a)
    lib.call = call
import bench
bench.main(sys.argv[2:])
