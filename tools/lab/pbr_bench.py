import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
N, S, D = 1280, 197, 768
dx = (torch.rand((N, S, D), device='cuda') - 0.5).to(torch.bfloat16)
dpos = torch.zeros((S, D), device='cuda'); dcls = torch.zeros(D, device='cuda'); db = torch.zeros(D, device='cuda')
for _ in range(3): ops.patch_embed_bwd_reduce(dx, dpos, dcls, db, N, S, D)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.patch_embed_bwd_reduce(dx, dpos, dcls, db, N, S, D)
e1.record(); torch.cuda.synchronize()
print('patch_bwd_reduce %.1f us' % (e0.elapsed_time(e1) * 100))
dpos.zero_(); dcls.zero_(); db.zero_(); ops.patch_embed_bwd_reduce(dx, dpos, dcls, db, N, S, D); torch.cuda.synchronize()
ref = dx.float().sum(0)
print('err', float((dpos - ref).abs().max() / ref.abs().max()), float((dcls - ref[0]).abs().max()), float((db - ref[1:].sum(0)).abs().max() / ref[1:].sum(0).abs().max()))
