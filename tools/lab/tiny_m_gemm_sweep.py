"""The AVT-h head's GEMMs at the reference's own batch (3 clips x 10 frames = 30 rows; also 8 clips): a handful of 64 x 64 tiles, each streaming a long
reduction -- us per launch for the 2- / 3- / 4- / 6-deep rings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.05).to(torch.bfloat16)
def t(fn, it=50):
    try:
        for _ in range(5): fn()
    except Exception as e:
        return None
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
TILES = (0, 64, 643, 644, 646, 128)
for M in (30, 80, 160):
    print(f'== M = {M}', flush=True)
    for name, K, N in [('c_proj 2048->2048', 2048, 2048), ('mlp.c_proj 8192->2048', 8192, 2048), ('c_attn 2048->6144', 2048, 6144), ('mlp.c_fc 2048->8192', 2048, 8192),
                       ('classifier 2048->3840', 2048, 3840)]:
        x = r(M, K); w_io = r(K, N); w_oi = r(N, K)
        bias = torch.randn(N, device='cuda', generator=g)
        for lay, fn in [('fwd  (B [K][N])', lambda tile: ops.gemm(x, w_io, M, N, K, a_kmajor=True, b_kmajor=False, bias=bias, tile=tile)),
                        ('dgrad (k-major)', lambda tile: ops.gemm(x, w_oi, M, N, K, a_kmajor=True, b_kmajor=True, tile=tile))]:
            row = []
            for tile in TILES:
                us = t(lambda: fn(tile))
                row.append(f'{tile}: ' + ('   n/a' if us is None else f'{us:6.1f}'))
            print(f'{name:24s} {lay:16s} ' + '  '.join(row) + f'   (weights {N * K * 2 / 1e6:.1f} MB)', flush=True)
