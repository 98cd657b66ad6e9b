"""Tile choice of the ViT GEMMs at small batches (3 / 8 / 16 clips per GPU: M = 5910 / 15760 / 31520 token rows): us per launch for every tile the router can take."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
D = 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(fn, iters=30):
    try:
        for _ in range(3): fn()
    except Exception as e:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
TILES = (0, 64, 643, 128, 256, 808)
for clips in (3, 8, 16):
    M = clips * 10 * 197
    print(f'== {clips} clips: M = {M}', flush=True)
    for name, N, K, kind in [('proj fwd +res', D, D, 'res'), ('fc2 fwd +res', D, 4 * D, 'res'), ('qkv dgrad', D, 3 * D, 'plain'), ('fc1 dgrad', D, 4 * D, 'plain'),
                             ('qkv fwd', 3 * D, D, 'plain'), ('fc1 fwd GELU + GELU\'', 4 * D, D, 'gelu'), ('fc2 dgrad x aux + colsum', 4 * D, D, 'aux')]:
        a, b = r(M, K), r(N, K)
        bias = torch.rand(N, device='cuda')
        out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
        kw = {}
        if kind == 'res': kw = dict(bias=bias, res=r(M, N))
        elif kind == 'plain': kw = dict(bias=bias)
        elif kind == 'gelu': kw = dict(bias=bias, act=ops.ACT_GELU_ERF, c2=torch.empty_like(out))
        elif kind == 'aux': kw = dict(act=ops.ACT_MUL_AUX, aux=r(M, N), colsum=torch.zeros(N, device='cuda'))
        res = []
        for t in TILES:
            us = timeit(lambda: ops.gemm(a, b, M, N, K, out=out, tile=t, **kw))
            res.append('     -' if us is None else f'{us:6.1f}')
        print(f'{name:28s} N {N:5d} K {K:5d}  ' + '  '.join(f'{t}: {x}' for t, x in zip(TILES, res)), flush=True)
