"""Race screen: the 8-phase GEMM (counted vmcnt waits, ping-pong groups) and the persistent attention kernels must be
bit-reproducible call after call; a rare early read of a staged buffer would show up as a differing output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for name, M, N, K, kw in [('fc2 fwd NT', 63040, 768, 3072, dict()), ('qkv fwd NT', 63040, 2304, 768, dict()),
                          ('fc1 dgrad NN', 63040, 768, 3072, dict(b_kmajor=False)), ('proj dgrad NN', 63040, 768, 768, dict(b_kmajor=False))]:
    a = r((M, K)); b = r((N, K)) if kw.get('b_kmajor', True) else r((K, N))
    bias = torch.rand(N, device='cuda'); res = r((M, N))
    ref = ops.gemm(a, b, M, N, K, bias=bias, res=res, **kw).clone()
    ref_f = (a.float() @ (b.float().t() if kw.get('b_kmajor', True) else b.float()) + bias + res.float())
    err = float((ref.float() - ref_f).abs().max() / ref_f.abs().max())
    nd = 0
    for _ in range(iters):
        out = ops.gemm(a, b, M, N, K, bias=bias, res=res, **kw)
        if not torch.equal(out, ref): nd += 1
    print(f'{name:14s}: {iters} calls, {nd} differ from the first; rel err vs fp32 {err:.2e}')
    bad += nd
frames, S, H = 640, 197, 12
qkv = r((frames * S, 3 * H * 64))
o0, l0 = ops.vit_attn_fwd(qkv, frames, S, H); o0 = o0.clone(); l0 = l0.clone()
d0 = ops.vit_attn_bwd(qkv, o0, o0, l0, frames, S, H).clone()
nf = nb = 0
for _ in range(max(iters // 4, 5)):
    o, l = ops.vit_attn_fwd(qkv, frames, S, H)
    if not (torch.equal(o, o0) and torch.equal(l, l0)): nf += 1
    d = ops.vit_attn_bwd(qkv, o0, o0, l0, frames, S, H)
    if not torch.equal(d, d0): nb += 1
print(f'attention fwd/bwd: {max(iters // 4, 5)} calls each, {nf} / {nb} differ from the first')
print('RACE SCREEN', 'CLEAN' if bad + nf + nb == 0 else 'FAILED')
