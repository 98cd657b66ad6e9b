"""How fast is an HBM-bound kernel (the folded LayerNorm backward, 3.1 GB) when only some CUs are free?  `held` CUs are occupied by sleeping workgroups
that take a CU's whole LDS (tools/lab/cu_hog159.hip) on a second stream."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from avt_amd import ops
hog = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libcu_hog159.so'))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device='cuda', dtype=torch.int32)
side = torch.cuda.Stream()
M, D = 2560 * 197, 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x, dres, dln = r(M, D), r(M, D), r(M, D)
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous()
cs = torch.zeros(D, device='cuda')
for held in (0, 128, 192, 216, 232, 240):
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize()
        ops.layernorm_bwd_folded(dln, x, sf, dres=dres, colsum=cs)
        if held:
            hog.cu_hog(held, 20000, sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(400000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.layernorm_bwd_folded(dln, x, sf, dres=dres, colsum=cs); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    print(f'{held:3d} CUs held, {256 - held:3d} free: ln_bwd_folded {best:8.1f} us = {M * D * 8 / best / 1e6:5.2f} TB/s', flush=True)
