"""Lab: the AVT-h head's GEMMs (M = 2560 rows at 256 clips x 10 frames; Conv1D weights stored (in, out)) under every tile choice.
usage: python tools/lab/head_gemm_sweep.py"""
import sys, torch
sys.path.insert(0, '.')
from avt_amd import ops
M = 2560
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.05).to(torch.bfloat16)
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
for name, K, N in [('c_proj 2048->2048', 2048, 2048), ('mlp.c_proj 8192->2048', 8192, 2048), ('c_attn 2048->6144', 2048, 6144), ('mlp.c_fc 2048->8192', 2048, 8192)]:
    x = r(M, K)
    w_io = r(K, N)            # Conv1D weight (in, out): forward reads it reduction-index-major
    w_oi = r(N, K)            # the same contraction with a k-major weight (data gradient of the transposed layer)
    bias = torch.randn(N, device='cuda', generator=g)
    for lay, fn in [('fwd  (A k-major, B [K][N])', lambda tile: ops.gemm(x, w_io, M, N, K, a_kmajor=True, b_kmajor=False, bias=bias, tile=tile)),
                    ('dgrad (both k-major)', lambda tile: ops.gemm(x, w_oi, M, N, K, a_kmajor=True, b_kmajor=True, tile=tile))]:
        row = []
        for tile in (0, 64, 643, 128, 256, 808):
            try:
                us = min(t(lambda: fn(tile)) for _ in range(2))
                row.append(f'{tile}: {us:6.1f}')
            except Exception as e:
                row.append(f'{tile}: n/a')
        print(f'{name:24s} {lay:28s} ' + '   '.join(row) + f'   ({2.0 * M * N * K / 1e9:.1f} GF)', flush=True)
