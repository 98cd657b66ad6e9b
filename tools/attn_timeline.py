import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
frames, S, H = 320, 197, 12
dbg = torch.zeros(frames * H * 16 * 4, device='cuda', dtype=torch.int64)
os.environ['AVT_ATTN_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
qkv = (torch.rand((frames * S, 3 * H * 64), device='cuda') * 2 - 1).to(torch.bfloat16)
o, lse = ops.vit_attn_fwd(qkv, frames, S, H)
for _ in range(2):
    ops.vit_attn_bwd(qkv, o, o, lse, frames, S, H); torch.cuda.synchronize()
d = dbg.view(frames * H, 16, 4)[:, :13].double()
print('per wave avg cycles: staging+D %.0f  phaseA %.0f  phaseB+store %.0f' % (d[..., 0].mean(), d[..., 1].mean(), d[..., 2].mean()))
print('per block max over waves: staging %.0f  A %.0f  B %.0f' % (d[..., 0].max(1).values.mean(), d[..., 1].max(1).values.mean(), d[..., 2].max(1).values.mean()))
