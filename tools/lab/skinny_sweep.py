"""GEMMs with at most 32 output rows (the head at the reference's 3 clips per GPU: 30 rows; the CLS-only last ViT block): us per launch and weight-stream
rate of the skinny kernel (tile 32) against the 64 x 64 kernels (643 = 3-deep ring, 64 = 2-deep) and 128 x 128.   usage: python tools/lab/skinny_sweep.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
shapes = [('head c_attn  fwd', 6144, 2048, False), ('head c_proj  fwd', 2048, 2048, False), ('head c_fc    fwd', 8192, 2048, False), ('head mlp.proj fwd', 2048, 8192, False),
          ('head c_attn  dgrad', 2048, 6144, True), ('head c_proj  dgrad', 2048, 2048, True), ('head c_fc    dgrad', 2048, 8192, True), ('head mlp.proj dgrad', 8192, 2048, True),
          ('cls-block proj', 768, 768, True), ('cls-block fc1', 3072, 768, True), ('cls-block fc2', 768, 3072, True), ('cls-block q', 768, 768, True)]
print(f'# M = {M} rows; us per launch (HIP events over 200 launches of 8 rotating weight copies -- every launch streams its weights from HBM), weight bytes / time')
for name, N, K, kk in shapes:
    a = r(M, K)
    ws = [r(N, K) if kk else r(K, N) for _ in range(8)]
    row = f'{name:20s} N {N:5d} K {K:5d} {"NT" if kk else "NN"}  '
    for tile in (32, 643, 64, 128):
        if tile == 32 and M > 64:
            continue
        for w in ws:
            ops.gemm(a, w, M, N, K, a_kmajor=True, b_kmajor=kk, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            ops.gemm(a, ws[i % 8], M, N, K, a_kmajor=True, b_kmajor=kk, tile=tile)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        row += f'  tile {tile:3d}: {us:6.1f} us {N * K * 2 / us / 1e6:5.2f} TB/s'
    print(row, flush=True)
