"""How fast is the vendor library (torch.matmul -> hipBLASLt) on the step's GEMM shapes?  Headroom probe only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
shapes = [('8192^3', 8192, 8192, 8192), ('qkv fwd', 63040, 2304, 768), ('proj fwd', 63040, 768, 768),
          ('fc1 fwd', 63040, 3072, 768), ('fc2 fwd', 63040, 768, 3072)]
for name, M, N, K in shapes:
    x, w = r((M, K)), r((N, K))
    t_lib = timeit(lambda: torch.matmul(x, w.t()))
    t_own = timeit(lambda: ops.gemm(x, w, M, N, K))
    fl = 2.0 * M * N * K
    print(f'{name:9s} NT  hipBLASLt {fl/t_lib/1e12:7.1f} TF/s   avt_gemm_bf16 {fl/t_own/1e12:7.1f} TF/s')
# wgrad shape: dW[N,K] = dY^T[N,M] X[M,K]
for name, M, N, K in [('fc1 wgrad', 63040, 3072, 768), ('qkv wgrad', 63040, 2304, 768)]:
    dy, x = r((M, N)), r((M, K))
    t_lib = timeit(lambda: torch.matmul(dy.t(), x))
    fl = 2.0 * M * N * K
    print(f'{name:9s} TN  hipBLASLt {fl/t_lib/1e12:7.1f} TF/s')
