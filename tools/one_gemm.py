"""Run one GEMM configuration a few times (for rocprofv3 --pmc passes). usage: one_gemm.py M N K layout tile [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avt_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
layout, tile = sys.argv[4], int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
if layout == 'NT':
    a, b = r((M, K)), r((N, K)); f = lambda: ops.gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=True, tile=tile)
elif layout == 'NN':
    a, b = r((M, K)), r((K, N)); f = lambda: ops.gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=False, tile=tile)
else:
    a, b = r((K, M)), r((K, N)); c = torch.zeros((M, N), device='cuda')
    f = lambda: ops.gemm(a, b, M, N, K, a_kmajor=False, b_kmajor=False, out=c, out_mode=2, tile=tile, splitk=int(os.environ.get('SPLITK', '0')))
for _ in range(iters): f()
torch.cuda.synchronize()
