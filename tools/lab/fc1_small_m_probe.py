"""fc1 forward at the reference's own batch (M = 5910 rows): the folded / unfolded GELU epilogue of the one-tile 8-phase kernel in isolation, warm and with the
operands evicted between calls (a 600-MB fill in between), to explain the 90 us it takes inside the 3-clip step against 53 us in a warm loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
M, D = 5910, 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x, W = r(M, D), r(4 * D, D)
b, c = torch.rand(4 * D, device='cuda'), torch.rand(4 * D, device='cuda')
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous()
o, d = torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16), torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16)
junk = torch.empty(300 << 20, device='cuda', dtype=torch.bfloat16)
def timeit(name, fn, evict, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if evict: junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f'{name:60s} {"evicted" if evict else "warm   "} {tot * 1e3 / iters:8.1f} us', flush=True)
for ev in (False, True):
    timeit('fc1 fwd GELU + GELU\' (unfolded)', lambda: ops.linear_fwd(x, W, bias=b, act=ops.ACT_GELU_ERF, c2=d, out=o), ev)
    timeit('fc1 fwd GELU + GELU\' folded', lambda: ops.linear_fwd(x, W, bias=b, act=ops.ACT_GELU_ERF, c2=d, out=o, ln_stat=sf, ln_c=c), ev)
    timeit('fc1 fwd GELU only (no second output)', lambda: ops.linear_fwd(x, W, bias=b, act=ops.ACT_GELU_ERF, out=o), ev)
    timeit('plain bias, N = 3072', lambda: ops.linear_fwd(x, W, bias=b, out=o), ev)
    timeit('plain bias, N = 2304 (qkv)', lambda: ops.linear_fwd(x, W[:2304], bias=b[:2304], out=o[:, :2304]), ev)
