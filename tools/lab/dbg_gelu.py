import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
M, N, K = 20480, 768, 64
g = torch.Generator().manual_seed(77)
a = (torch.randn((M, K), generator=g) * 0.6).to(torch.bfloat16).cuda()
b = (torch.randn((N, K), generator=g) * 0.25).to(torch.bfloat16).cuda()
bias = torch.zeros(N, dtype=torch.float32)
bias[0:8] = torch.tensor([256., -256., 5000., -7e4, 1e-6, -3e-6, 0., 300.])
bias[8] = float('inf'); bias[9] = float('-inf'); bias[10] = float('nan')
bias = bias.cuda()
h = torch.empty((M, N), device='cuda', dtype=torch.bfloat16); d = torch.empty_like(h)
ops.gemm(a, b, M, N, K, bias=bias, act=ops.ACT_GELU_ERF, c2=d, out=h)
plain = ops.gemm(a, b, M, N, K, bias=bias)
torch.cuda.synchronize()
pb = plain.float()
pos = pb >= 256.0
neq = pos & (h.float() != pb)
print('pos', int(pos.sum()), 'mismatch', int(neq.sum()))
idx = neq.nonzero()[:12]
for r, c in idx.tolist():
    print(r, c, float(pb[r, c]), float(h[r, c]), float(d[r, c]))
print('cols with mismatches:', sorted(set(neq.nonzero()[:, 1].tolist()))[:20])
print('rows mod 32 with mismatches:', sorted(set((neq.nonzero()[:, 0] % 32).tolist())))
