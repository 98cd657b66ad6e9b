import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avt_amd import ops
frames, S, H = 1, 208, 1
D = H * 64
g = torch.Generator(device='cuda').manual_seed(30)
qkv = (torch.randn((frames * S, 3 * D), device='cuda', generator=g)).to(torch.bfloat16)
out, lse = ops.vit_attn_fwd(qkv, frames, S, H)
dout = (torch.randn((frames * S, D), device='cuda', generator=g)).to(torch.bfloat16)
dqkv = ops.vit_attn_bwd(qkv, out, dout, lse, frames, S, H)
torch.cuda.synchronize()
dv = dqkv[:, 2 * D:].float()
bad = ~torch.isfinite(dv)
print('nonfinite count', int(bad.sum()), 'rows', bad.any(1).nonzero().flatten().tolist()[:40], 'cols', bad.any(0).nonzero().flatten().tolist()[:70])
print('lse finite', bool(torch.isfinite(lse).all()), float(lse.min()), float(lse.max()))
t = qkv.float().view(frames, S, 3, H, 64).permute(2, 0, 3, 1, 4)
q, k, v = t[0], t[1], t[2]
att = (q @ k.transpose(-2, -1)) * 0.125
print('max score', float(att.max()), 'min lse ref', float(torch.logsumexp(att, -1).min()))
