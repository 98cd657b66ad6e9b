"""Do the weight-gradient GEMMs of a ViT block's MLP backward, issued on a second stream, fill the tails of the data-gradient chain (and vice versa)?
One block's MLP backward at the step's size (256 clips): fc2 wgrad, fc2 dgrad (x GELU', scaled), fc1 wgrad, fc1 dgrad, folded LayerNorm backward --
back to back on one stream against the two weight gradients on a side stream (same kernels, same arguments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
M, D = 2560 * 197, 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
dx, x1, act = r(M, D), r(M, D), r(M, 4 * D)
W2t, G2t = r(4 * D, D), r(D, 4 * D)
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous(); sb = torch.stack([rstd, 1 / rstd], 1).contiguous()
pre = ops.FragTensor(M, 4 * D, dx.device) if ops.gemm_frag_ok(M, 4 * D, D) else r(M, 4 * D)
if type(pre) is ops.FragTensor: pre.buf.copy_(r(pre.buf.numel()))
dW2, T1 = torch.zeros((D, 4 * D), device='cuda'), torch.zeros((4 * D, D), device='cuda')
dbt, cs = torch.zeros(4 * D, device='cuda'), torch.zeros(D, device='cuda')
dh = torch.empty((M, 4 * D), device='cuda', dtype=torch.bfloat16); dln = torch.empty((M, D), device='cuda', dtype=torch.bfloat16)
main = torch.cuda.current_stream()
def timeit(name, fn, iters=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:64s} {e0.elapsed_time(e1) * 1e3 / iters:9.1f} us', flush=True)
def one_stream():
    ops.linear_wgrad(dx, act, dW2)
    ops.linear_fwd(dx, W2t, act=ops.ACT_MUL_AUX, aux=pre, colsum=dbt, ln_stat=sb, out=dh)
    ops.linear_wgrad(dh, x1, T1)
    ops.linear_fwd(dh, G2t, out=dln)
    ops.layernorm_bwd_folded(dln, x1, sf, dres=dx, colsum=cs)
def two_streams(side):
    e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
    with torch.cuda.stream(side):
        ops.linear_wgrad(dx, act, dW2)
    ops.linear_fwd(dx, W2t, act=ops.ACT_MUL_AUX, aux=pre, colsum=dbt, ln_stat=sb, out=dh)
    e1 = torch.cuda.Event(); e1.record(main); side.wait_event(e1)
    with torch.cuda.stream(side):
        ops.linear_wgrad(dh, x1, T1)
    ops.linear_fwd(dh, G2t, out=dln)
    ops.layernorm_bwd_folded(dln, x1, sf, dres=dx, colsum=cs)
    e2 = torch.cuda.Event(); e2.record(side); main.wait_event(e2)
for rep in range(2):
    timeit('one stream', one_stream)
    for prio, nm in ((0, 'same priority'), (-1, 'high priority'),):
        side = torch.cuda.Stream(priority=prio)
        timeit(f'weight gradients on a side stream ({nm})', lambda: two_streams(side))
