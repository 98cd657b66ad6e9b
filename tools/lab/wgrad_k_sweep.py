"""Lab: weight-gradient GEMM (TN, split-K, fp32 atomics) rate vs the reduction length -- do the operands' residency (L2 / MALL / HBM) matter?"""
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
N, Kd = 3072, 768
dw = torch.zeros((N, Kd), device='cuda')
for M in (8192, 16384, 32768, 65536, 126080, 252160):
    dy, x = r((M, N)), r((M, Kd))
    t = timeit(lambda: ops.linear_wgrad(dy, x, dw))
    print(f'rows {M:7d}: operands {(M*N+M*Kd)*2/1e6:7.0f} MB  {2.0*M*N*Kd/t/1e12:7.1f} TF/s  {t*1e6:8.1f} us')
