"""Weight gradients at small batches (3 / 8 / 16 clips per GPU: the reduction runs over 5910 / 15760 / 31520 token rows): us per call of avt_gemm_accum_bf16
(split-K slabs + the ordered reduction, both kernels) for the automatic choice (the 4-wave 256 x 256 kernel), 128 x 128 tiles and the 8-phase kernel."""
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
TILES = (0, 128, 808)
for clips in (3, 8, 16, 32):
    M = clips * 10 * 197
    print(f'== {clips} clips: reduction over M = {M} rows', flush=True)
    for name, N, K in [('proj wgrad', D, D), ('qkv wgrad', 3 * D, D), ('fc1 wgrad', 4 * D, D), ('fc2 wgrad', D, 4 * D)]:
        dy, x = r(M, N), r(M, K)
        dw = torch.zeros((N, K), device='cuda')
        res = []
        for t in TILES:
            ops.WGRAD_TILE = t
            us = timeit(lambda: ops.linear_wgrad(dy, x, dw))
            res.append('     -' if us is None else f'{us:6.1f}')
        ops.WGRAD_TILE = 0
        print(f'{name:12s} dW {N:5d} x {K:5d}  ' + '  '.join(f'{t}: {x_}' for t, x_ in zip(TILES, res)), flush=True)
