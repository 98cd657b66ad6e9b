"""Lab: which part of the attention bias gradient is not bit-reproducible?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import ops
def rnd(shape, s, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(shape, device='cuda', generator=g) * s).to(torch.bfloat16)
for det in (True,):
    ops.DETERMINISTIC_REDUCTIONS = det
    for frames, S, H in [(300, 64, 4), (300, 50, 4), (300, 33, 4), (300, 20, 4), (300, 32, 4), (300, 100, 4), (300, 128, 4), (300, 150, 4)]:
        qkv = rnd((frames * S, 3 * H * 64), 1.0, 47)
        o, l = ops.vit_attn_fwd(qkv, frames, S, H)
        do = rnd((frames * S, H * 64), 1.0, 48)
        outs = []
        for i in range(3):
            dbias = torch.zeros(3 * H * 64, device='cuda')
            dqkv = ops.vit_attn_bwd(qkv, o, do, l, frames, S, H, dbias=dbias)
            outs.append((dbias.clone(), dqkv.clone()))
        torch.cuda.synchronize()
        D = H * 64
        ref = outs[0]
        for i, (b, d) in enumerate(outs[1:]):
            parts = [int((b[j * D:(j + 1) * D] != ref[0][j * D:(j + 1) * D]).sum()) for j in range(3)]
            print(det, (frames, S, H), i, 'dbias mismatches q/k/v', parts, 'dqkv equal', bool(torch.equal(d, ref[1])), 'max|dk part|', float(b[D:2 * D].abs().max()))
        col = outs[0][1].double().sum(0).float()
        print('   vs colsum(dqkv): q %.2e  v %.2e' % (float((ref[0][:D] - col[:D]).abs().max() / col[:D].abs().max()), float((ref[0][2 * D:] - col[2 * D:]).abs().max() / col[2 * D:].abs().max())))
