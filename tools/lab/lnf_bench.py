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
# the folded forward GEMMs next to the plain ones (same shapes)
W3, W4 = r(3 * D, D), r(4 * D, D)
b3, b4 = torch.rand(3 * D, device='cuda'), torch.rand(4 * D, device='cuda')
c3, c4 = torch.rand(3 * D, device='cuda'), torch.rand(4 * D, device='cuda')
o3 = torch.empty((M, 3 * D), device='cuda', dtype=torch.bfloat16)
o4, d4 = torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16), torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16)
timeit('qkv fwd plain (EPK 0)', lambda: ops.linear_fwd(x, W3, bias=b3, out=o3))
timeit('qkv fwd folded (EPK 5)', lambda: ops.linear_fwd(x, W3, bias=b3, out=o3, ln_stat=sf, ln_c=c3))
timeit('fc1 fwd GELU plain (EPK 1)', lambda: ops.linear_fwd(x, W4, bias=b4, act=ops.ACT_GELU_ERF, c2=d4, out=o4))
timeit('fc1 fwd GELU folded (EPK 6)', lambda: ops.linear_fwd(x, W4, bias=b4, act=ops.ACT_GELU_ERF, c2=d4, out=o4, ln_stat=sf, ln_c=c4))
res = r(M, D); y768 = torch.empty((M, D), device='cuda', dtype=torch.bfloat16); Wp = r(D, D); bp = torch.rand(D, device='cuda')
timeit('proj fwd +res (EPK 2)', lambda: ops.linear_fwd(x, Wp, bias=bp, res=res, out=y768))
timeit('proj fwd +res +stats (EPK 4)', lambda: ops.linear_fwd(x, Wp, bias=bp, res=res, out=y768, stat_part=part))
W2t = r(4 * D, D)
timeit('fc2 dgrad x aux +colsum (EPK 3)', lambda: ops.linear_fwd(x, W2t, act=ops.ACT_MUL_AUX, aux=d4, colsum=b4, out=o4))
timeit('fc2 dgrad x aux +colsum scaled (EPK 7)', lambda: ops.linear_fwd(x, W2t, act=ops.ACT_MUL_AUX, aux=d4, colsum=b4, out=o4, ln_stat=sb))
del o3, o4, d4, res, y768
qkv = r(M, 3 * D)
out, lse = ops.vit_attn_fwd(qkv, N, S, H)
do = r(M, D); dbias = torch.zeros(3 * D, device='cuda')
timeit('vit_attn_bwd', lambda: ops.vit_attn_bwd(qkv, out, do, lse, N, S, H, dbias=dbias))
timeit('vit_attn_bwd scaled', lambda: ops.vit_attn_bwd(qkv, out, do, lse, N, S, H, dbias=dbias, row_stat=sb))
