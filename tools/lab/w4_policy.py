"""The weight-gradient kernel (gemm_w4_kernel) on the step's four shapes at 256 clips: us per launch (HIP events) -- and, run under
rocprofv3 --pmc FETCH_SIZE, its fabric reads per launch.  usage: AVT_HIP_LIB=... python tools/lab/w4_policy.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, D = 2560 * 197, 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x, dh, dqkv = r(M, D), r(M, 4 * D), r(M, 3 * D)
for name, dy, xin, n in [('fc1 wgrad', dh, x, 4 * D), ('fc2 wgrad', x, dh, D), ('qkv wgrad', dqkv, x, 3 * D), ('proj wgrad', x, x, D)]:
    dw = torch.zeros((n, xin.size(1)), device='cuda')
    for _ in range(3):
        ops.linear_wgrad(dy, xin, dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear_wgrad(dy, xin, dw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f'{name:12s} {us:9.1f} us  {2.0 * M * n * xin.size(1) / us / 1e6:7.1f} TF/s  operands {(dy.numel() + xin.numel()) * 2 / 1e9:.2f} GB', flush=True)
