"""Lab: effective shader clock of a GEMM launch = (in-kernel cycle-counter span of the launch, per XCD) / (event time of the launch), for
gemm_8p_kernel and the persistent kernel on the same shapes: does the kernel that keeps the matrix pipe busier run at a lower clock?
usage: python tools/lab/persist_clock.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(ROOT, 'avt_amd', 'libavt_hip_lab.so'))
dbg = torch.zeros(256 * 128 * 2 * 8, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
M = 2560 * 197
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x768, x3072 = r(M, 768), r(M, 3072)
cases = [('qkv fwd K=768 N=2304', x768, 2304), ('fc1 dgrad K=3072 N=768', x3072, 768), ('qkv dgrad-like K=3072 N=2304', x3072, 2304)]
for name, A, N in cases:
    K = A.size(1)
    W = r(N, K)
    out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
    for mode in ('0', '15', '0', '15'):
        os.environ['AVT_GEMM_PERSIST'] = mode
        for _ in range(3):
            ops.linear_fwd(A, W, out=out)
        torch.cuda.synchronize()
        dbg.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear_fwd(A, W, out=out); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        d = dbg.view(-1, 8); d = d[d[:, 0] > 0].cpu().double()
        xcc = d[:, 5].long() & 7
        spans = []
        for x in range(8):
            dx = d[xcc == x]
            if len(dx):
                spans.append(float(dx[:, 3].max() - dx[:, 0].min()))
        span = sum(spans) / len(spans)
        print(f'{name:30s} {"persistent" if mode != "0" else "8p        "}  {us:8.1f} us   span {span / 1e6:6.3f} M cycles   -> {span / us / 1e3:5.3f} GHz   records {len(d)}', flush=True)
