"""Lab: where do the waves of the ping-pong GEMM spend their cycles (vmcnt wait / barrier / compute / LDS reads)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'avt_amd', 'libavt_hip_lab.so'))   # lab build: make -C avt_amd/csrc lab
dbg = torch.zeros(16 * 20000, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
for name, M, N, K in [('fc2 fwd', 63040, 768, 3072), ('fc1 fwd', 63040, 3072, 768), ('8192^3', 8192, 8192, 8192)]:
    x, w = r((M, K)), r((N, K))
    for _ in range(2):
        dbg.zero_(); ops.gemm(x, w, M, N, K, tile=512); torch.cuda.synchronize()
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    d = dbg[:16 * nb].view(nb, 2, 8).double()
    for g in (0, 1):
        m = d[:, g].mean(0)
        nk = int(m[6])
        print(f'{name:8s} G{g}: per K tile: vmcnt-wait {m[0]/nk:7.0f}  barrier {m[1]/nk:7.0f}  compute {m[2]/nk:7.0f}  lds-read {m[3]/nk:7.0f}  dma-issue {m[7]/nk:7.0f} | loop total {m[4]/nk:7.0f}/tile  epilogue {m[5]:8.0f}')
