"""4-wave k-major GEMM (gemm_k4.hip, tile 2566) against gemm_8p_kernel (tile 808): bit equality (incl. guard rows behind a ragged M) and time.
usage: python tools/lab/k4_check.py [frames=2560]
(Historical: tile 2566 exists only when tools/lab/attic/gemm_k4*.hip.txt is put back as avt_amd/csrc/gemm_k4.hip with its dispatch hook -- the kernels
measured slower and were removed; profiles/r04_persistent_gemm.txt section 6.)"""
import os, sys
import torch
sys.path.insert(0, '.')
from avt_amd import ops
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
M = frames * 197
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *shape, scale=1.0: (torch.randn(shape, device=dev, generator=g) * scale).to(torch.bfloat16)

def timed(fn, it=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

x3072, x2304, x768 = rnd(M, 3072, scale=0.5), rnd(M, 2304, scale=0.5), rnd(M, 768)
resid, pre = rnd(M, 768), rnd(M, 3072)
cases = [('fc2 fwd   K=3072 N=768 bias+res', x3072, 768, dict(bias=True, res=True)),
         ('fc1 dgrad K=3072 N=768', x3072, 768, dict()),
         ('qkv dgrad K=2304 N=768', x2304, 768, dict()),
         ('qkv fwd   K=768  N=2304 bias', x768, 2304, dict(bias=True)),
         ('fc2 dgrad K=768  N=3072 *aux colsum', x768, 3072, dict(act=ops.ACT_MUL_AUX, aux=True, colsum=True))]
ok = True
for name, A, N, kw in cases:
    K = A.size(1)
    W = rnd(N, K, scale=0.03)
    args = {}
    if kw.get('bias'): args['bias'] = torch.randn(N, device=dev, generator=g)
    if 'act' in kw: args['act'] = kw['act']
    if kw.get('res'): args['res'] = resid
    if kw.get('aux'): args['aux'] = pre
    outs, best = {}, {808: 1e30, 2566: 1e30}
    for r_ in range(4):
        for tile in (808, 2566):
            full = torch.full((M + 256, N), 7.0, device=dev, dtype=torch.bfloat16)
            cs = torch.zeros(N, device=dev) if kw.get('colsum') else None
            a = dict(args)
            if cs is not None: a['colsum'] = cs
            ops.gemm(A, W, M, N, K, out=full[:M], tile=tile, **a)
            torch.cuda.synchronize()
            outs[tile] = (full, cs)
            best[tile] = min(best[tile], timed(lambda: ops.gemm(A, W, M, N, K, out=full[:M], tile=tile, **args)))
    same = torch.equal(outs[808][0].view(torch.int16), outs[2566][0].view(torch.int16)) and (outs[808][1] is None or torch.equal(outs[808][1], outs[2566][1]))
    ok = ok and same
    tf = 2.0 * M * N * K / 1e12
    print(f'{name:38s} 8p {best[808]:8.1f} us ({tf / best[808] * 1e3:5.3f} PF/s)   4-wave {best[2566]:8.1f} us ({tf / best[2566] * 1e3:5.3f} PF/s) {(best[808] / best[2566] - 1) * 100:+5.1f} %   bit-equal {same}', flush=True)
print('ALL BIT-EQUAL' if ok else 'MISMATCH', flush=True)
