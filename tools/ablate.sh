#!/bin/bash
export AVT_HIP_LIB=${AVT_HIP_LIB:-$(pwd)/avt_amd/libavt_hip_lab.so}   # lab build (make -C avt_amd/csrc lab): the product library has no ablation / stagger switches
for a in 0 1 2 3 4; do echo "ABLATE=$a"; AVT_GEMM_ABLATE=$a python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from avt_amd import ops
n = 8192
a = (torch.rand((n, n), device='cuda') * 2 - 1).to(torch.bfloat16); b = a.clone()
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
t = bench(lambda: ops.gemm(a, b, n, n, n, tile=512)); print(f'  8192^3 NT pp: {2*n**3/t/1e12:7.1f} TF/s')
PY
done
