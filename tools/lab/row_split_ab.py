"""3 clips per GPU: the ViT's N = 3072 GEMMs (fc1 forward, fc2 data gradient) have 24 x 12 = 288 tiles of 256 x 256 = 1.125 rounds of the 256 CUs, paid as two.
A/B: the rows that fill one round (21 row tiles) on the 8-phase kernel, the remaining 534 rows on the 3-deep 64 x 64 ring, as two calls on row slices (same bits).
usage: python tools/lab/row_split_ab.py {split|one} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import ops
if sys.argv[1] == 'split':
    real = ops.gemm
    def rows(t, lo, hi):
        return None if t is None else t[lo:hi]
    def gemm(A, B, M, N, K, *, a_kmajor=True, b_kmajor=True, out=None, out_mode=ops.OUT_BF16, tile=0, res=None, aux=None, c2=None, res_period=0, **kw):
        tn, tm = (N + 255) // 256, (M + 255) // 256
        t256 = tm * tn
        if (tile == 0 and a_kmajor and b_kmajor and out_mode == ops.OUT_BF16 and res_period == 0 and 256 < t256 <= 320 and K % 64 == 0
                and type(aux) is not ops.FragTensor and type(c2) is not ops.FragTensor and kw.get('stat_part') is None and kw.get('ln_stat') is None):
            m0 = (256 // tn) * 256
            if 0 < m0 < M:
                import torch
                if out is None:
                    out = torch.empty((M, N), device=A.device, dtype=ops.BF16)
                real(A[:m0], B, m0, N, K, out=out[:m0], tile=808, res=rows(res, 0, m0), aux=rows(aux, 0, m0), c2=rows(c2, 0, m0), **kw)
                real(A[m0:], B, M - m0, N, K, out=out[m0:], tile=643, res=rows(res, m0, M), aux=rows(aux, m0, M), c2=rows(c2, m0, M), **kw)
                return out
        return real(A, B, M, N, K, a_kmajor=a_kmajor, b_kmajor=b_kmajor, out=out, out_mode=out_mode, tile=tile, res=res, aux=aux, c2=c2, res_period=res_period, **kw)
    ops.gemm = gemm
import bench
bench.main(sys.argv[2:])
