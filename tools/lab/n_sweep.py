"""Lab: forward GEMM rate vs N at fixed M, K (is N = 2304 special?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 63040, 768
x = r((M, K))
for N in (1536, 2048, 2304, 2560, 3072, 4096):
    w = r((N, K))
    t = timeit(lambda: ops.gemm(x, w, M, N, K))
    tiles = ((M + 255) // 256) * (N // 256)
    print(f'N={N:5d} tiles={tiles:6d} rounds={tiles/256:6.2f}  {2.0*M*N*K/t/1e12:7.1f} TF/s   per-round {t*1e6/ -(-tiles//256):6.1f} us')
