import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avt_amd import ops
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for n in (8192,):
    for fill in ('rand',):
        a = ((torch.rand((n, n), device='cuda') * 2 - 1) if fill == 'rand' else torch.zeros((n, n), device='cuda')).to(torch.bfloat16)
        b = a.clone()
        for tile in (256, 808):
            for lay, (ak, bk) in {'NT': (True, True), 'NN': (True, False)}.items():
                t = bench(lambda: ops.gemm(a, b, n, n, n, a_kmajor=ak, b_kmajor=bk, tile=tile))
                print(f'{n}^3 {fill} {lay} tile={tile}: {2*n**3/t/1e12:7.1f} TF/s')

M = 63040
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
for name, N, K in [('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    x, w = r((M, K)), r((N, K))
    for tile in (256, 808):
        t = bench(lambda: ops.gemm(x, w, M, N, K, tile=tile)); print(f'{name} plain NT tile={tile}: {2.0*M*N*K/t/1e12:7.1f} TF/s {t*1e6:7.1f} us')
