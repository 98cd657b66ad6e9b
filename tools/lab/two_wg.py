"""lab: two 4-wave workgroups per CU (tiles 2562 / 2563 / 1283 of libavt_hip_lab.so) against the 8-phase kernel on the step's
K = 768 GEMMs, with the odd threadgroup slot of the first dispatch wave started AVT_GEMM_STAGGER cycles late.
usage: AVT_HIP_LIB=.../libavt_hip_lab.so AVT_GEMM_STAGGER=N python tools/lab/two_wg.py tile [tile...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
B = int(os.environ.get('KB_BATCH', 256)); M = B * 10 * 197; D = 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(fn, iters=12, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
x, w1, w2t, wp, wq = r(M, D), r(3072, D), r(3072, D), r(D, D), r(3 * D, D)
b3, b1, bq = torch.rand(3072, device='cuda'), torch.rand(D, device='cuda'), torch.rand(3 * D, device='cuda')
pre = torch.empty((M, 3072), device='cuda', dtype=torch.bfloat16); act = torch.empty_like(pre)
y = torch.empty((M, D), device='cuda', dtype=torch.bfloat16); qkv = torch.empty((M, 3 * D), device='cuda', dtype=torch.bfloat16)
cs3 = torch.zeros(3072, device='cuda')
ref = None
for tile in [int(t) for t in sys.argv[1:]]:
    ops.FORCE_TILE = tile
    res = {}
    res['fc1 fwd gelu+gelu\''] = timeit(lambda: ops.linear_fwd(x, w1, bias=b3, act=ops.ACT_GELU_ERF, c2=pre, out=act))
    chk = act[:4096].float().clone()
    if ref is None: ref = chk
    err = float((chk - ref).abs().max())
    res['fc1 fwd plain'] = timeit(lambda: ops.linear_fwd(x, w1, out=act))
    res['fc2 dgrad x aux'] = timeit(lambda: ops.linear_fwd(x, w2t, act=ops.ACT_MUL_AUX, aux=pre, out=act))
    res['proj fwd +bias+res'] = timeit(lambda: ops.linear_fwd(x, wp, bias=b1, res=x, out=y))
    res['qkv fwd +bias'] = timeit(lambda: ops.linear_fwd(x, wq, bias=bq, out=qkv))
    res['fc2 fwd +bias+res (K=3072)'] = timeit(lambda: ops.linear_fwd(act, w2t.view(D, 3072), bias=b1, res=x, out=y))
    if tile != 0:          # bit-equality with the default kernel on every epilogue (same k order per output element)
        ops.FORCE_TILE = 0
        o0, c0 = torch.empty_like(act), torch.empty_like(pre)
        ops.linear_fwd(x, w1, bias=b3, act=ops.ACT_GELU_ERF, c2=c0, out=o0)
        ops.FORCE_TILE = tile
        o1, c1 = torch.empty_like(act), torch.empty_like(pre)
        ops.linear_fwd(x, w1, bias=b3, act=ops.ACT_GELU_ERF, c2=c1, out=o1)
        eq = [torch.equal(o0, o1), torch.equal(c0, c1)]
        ops.FORCE_TILE = 0; csA = torch.zeros(3072, device='cuda'); dA = ops.linear_fwd(x, w2t, act=ops.ACT_MUL_AUX, aux=c0, colsum=csA)
        ops.FORCE_TILE = tile; csB = torch.zeros(3072, device='cuda'); dB = ops.linear_fwd(x, w2t, act=ops.ACT_MUL_AUX, aux=c0, colsum=csB)
        eq += [torch.equal(dA, dB), torch.equal(csA, csB)]
        ops.FORCE_TILE = 0; yA = ops.linear_fwd(x, wp, bias=b1, res=x)
        ops.FORCE_TILE = tile; yB = ops.linear_fwd(x, wp, bias=b1, res=x)
        eq += [torch.equal(yA, yB)]
        xs = x[:1000 * 197 - 3]                       # ragged M (not a multiple of 256), N = 2304
        ops.FORCE_TILE = 0; qA = ops.linear_fwd(xs, wq, bias=bq)
        ops.FORCE_TILE = tile; qB = ops.linear_fwd(xs, wq, bias=bq)
        eq += [torch.equal(qA, qB)]
        print('   bit-equal to the default kernel [gelu out, gelu\' out, x aux out, colsum, proj+res, ragged qkv]:', eq, flush=True)
        del o0, c0, o1, c1, dA, dB, yA, yB, qA, qB
        ops.FORCE_TILE = tile
        res['fc2 dgrad x aux + colsum'] = timeit(lambda: ops.linear_fwd(x, w2t, act=ops.ACT_MUL_AUX, aux=pre, colsum=cs3, out=act))
    else:
        res['fc2 dgrad x aux + colsum'] = timeit(lambda: ops.linear_fwd(x, w2t, act=ops.ACT_MUL_AUX, aux=pre, colsum=cs3, out=act))
    print(f'tile {tile:5d} stagger {os.environ.get("AVT_GEMM_STAGGER", "0"):>6s} maxdiff-vs-first {err:.3g}: ' + '  '.join(f'{k} {v:7.1f}' for k, v in res.items()), flush=True)
