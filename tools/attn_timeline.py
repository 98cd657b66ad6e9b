import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'avt_amd', 'libavt_hip_lab.so'))   # lab build: make -C avt_amd/csrc lab
frames, S, H = 320, 197, 12
dbg = torch.zeros(frames * H * 16 * 8, device='cuda', dtype=torch.int64)
os.environ['AVT_ATTN_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
qkv = (torch.rand((frames * S, 3 * H * 64), device='cuda') * 2 - 1).to(torch.bfloat16)
o, lse = ops.vit_attn_fwd(qkv, frames, S, H)
for _ in range(2):
    ops.vit_attn_bwd(qkv, o, o, lse, frames, S, H); torch.cuda.synchronize()
nb = min(frames * H, 256)
d = dbg.view(frames * H, 16, 8)[:nb, :13].double()
n = d[..., 3].clamp(min=1)
print('per item, per wave avg cycles: barrier1+strips %.0f  phaseA+barrier2 %.0f  phaseB+stores %.0f' % ((d[..., 0] / n).mean(), (d[..., 1] / n).mean(), (d[..., 2] / n).mean()))

# per wave (= per SIMD: wave w sits on SIMD w % 4): where the time goes and how long it waits at the two barriers
pw = (d / n[..., None]).mean(0)
print('wave simd  seg1(barrier1+prep)  wait@b1   seg2(phaseA+barrier2)  wait@b2   seg3(phaseB+stores)   busy = total - waits')
for w in range(13):
    tot = float(pw[w, 0] + pw[w, 1] + pw[w, 2])
    print('%3d  %3d   %10.0f %10.0f   %14.0f %10.0f   %14.0f   %10.0f' % (w, w % 4, pw[w, 0], pw[w, 4], pw[w, 1], pw[w, 5], pw[w, 2], tot - float(pw[w, 4] + pw[w, 5])))
