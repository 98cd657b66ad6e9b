"""us per launch of the persistent GEMM's epilogue forms at the bench's size, for the timing-only ablation builds (AVT_PK_ABL, gemm_persist.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
N, S, D = 2560, 197, 768
M = N * S
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(name, fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:44s} {e0.elapsed_time(e1) * 1e3 / iters:9.1f} us', flush=True)
x = r(M, D)
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous(); sb = torch.stack([rstd, 1 / rstd], 1).contiguous()
W3, W4 = r(3 * D, D), r(4 * D, D)
b3, b4 = torch.rand(3 * D, device='cuda'), torch.rand(4 * D, device='cuda')
c4 = torch.rand(4 * D, device='cuda')
o3 = torch.empty((M, 3 * D), device='cuda', dtype=torch.bfloat16)
o4, d4 = torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16), torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16)
timeit('qkv fwd plain (EPK 0), N = 2304', lambda: ops.linear_fwd(x, W3, bias=b3, out=o3))
timeit('plain (EPK 0), N = 3072', lambda: ops.linear_fwd(x, W4, bias=b4, out=o4))
timeit('fc1 fwd GELU, one output (EPK 1)', lambda: ops.linear_fwd(x, W4, bias=b4, act=ops.ACT_GELU_ERF, out=o4))
timeit('fc1 fwd GELU + GELU\' (EPK 1)', lambda: ops.linear_fwd(x, W4, bias=b4, act=ops.ACT_GELU_ERF, c2=d4, out=o4))
timeit('fc1 fwd GELU + GELU\' folded (EPK 6)', lambda: ops.linear_fwd(x, W4, bias=b4, act=ops.ACT_GELU_ERF, c2=d4, out=o4, ln_stat=sf, ln_c=c4))
W2t = r(4 * D, D)
timeit('fc2 dgrad x aux +colsum (EPK 3)', lambda: ops.linear_fwd(x, W2t, act=ops.ACT_MUL_AUX, aux=d4, colsum=b4, out=o4))
timeit('fc2 dgrad x aux +colsum scaled (EPK 7)', lambda: ops.linear_fwd(x, W2t, act=ops.ACT_MUL_AUX, aux=d4, colsum=b4, out=o4, ln_stat=sb))
