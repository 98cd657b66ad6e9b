"""us per launch of the folded-LayerNorm pieces at the bench's size (2560 frames x 197 tokens x 768) and of the attention backward with / without row scaling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
N, S, D, H = 2560, 197, 768, 12
M = N * S
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(name, fn, bytes_=None, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f'{name:40s} {us:9.1f} us' + (f'  {bytes_ / us / 1e6:6.2f} TB/s' if bytes_ else ''), flush=True)
x, dy, dres = r(M, D), r(M, D), r(M, D)
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous(); sb = torch.stack([rstd, 1 / rstd], 1).contiguous()
cs = torch.zeros(D, device='cuda')
timeit('ln_bwd_folded (+dres, colsum)', lambda: ops.layernorm_bwd_folded(dy, x, sf, dres=dres, colsum=cs), bytes_=M * D * 8)
g, b = torch.rand(D, device='cuda') + 0.5, torch.rand(D, device='cuda')
y, mean, rs = ops.layernorm_fwd(x, g, b, 1e-6)
dg, db = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
timeit('ln_bwd (unfolded, +dres, colsum)', lambda: ops.layernorm_bwd(dy, x, mean, rs, g, dg, db, dres=dres, colsum=cs), bytes_=M * D * 8)
part = ops.ln_stat_part(M, D, x.device)
timeit('ln_stats_finalize', lambda: ops.ln_stats_finalize(part, D, 1e-6), bytes_=M * (12 * 16 + 16))
del dy, dres, y
qkv = r(M, 3 * D)
out, lse = ops.vit_attn_fwd(qkv, N, S, H)
do = r(M, D); dbias = torch.zeros(3 * D, device='cuda')
timeit('vit_attn_bwd', lambda: ops.vit_attn_bwd(qkv, out, do, lse, N, S, H, dbias=dbias))
timeit('vit_attn_bwd scaled', lambda: ops.vit_attn_bwd(qkv, out, do, lse, N, S, H, dbias=dbias, row_stat=sb))
