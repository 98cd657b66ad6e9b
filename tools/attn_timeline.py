import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'avt_amd', 'libavt_hip_lab.so'))   # lab build: make -C avt_amd/csrc lab
frames, S, H = 320, 197, 12
dbg = torch.zeros(frames * H * 16 * 4, device='cuda', dtype=torch.int64)
os.environ['AVT_ATTN_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
qkv = (torch.rand((frames * S, 3 * H * 64), device='cuda') * 2 - 1).to(torch.bfloat16)
o, lse = ops.vit_attn_fwd(qkv, frames, S, H)
for _ in range(2):
    ops.vit_attn_bwd(qkv, o, o, lse, frames, S, H); torch.cuda.synchronize()
nb = min(frames * H, 256)
d = dbg.view(frames * H, 16, 4)[:nb, :13].double()
n = d[..., 3].clamp(min=1)
print('per item, per wave avg cycles: barrier1+strips %.0f  phaseA+barrier2 %.0f  phaseB+stores %.0f' % ((d[..., 0] / n).mean(), (d[..., 1] / n).mean(), (d[..., 2] / n).mean()))
