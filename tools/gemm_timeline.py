"""Lab: per-block prologue+loop / epilogue timing of the generic GEMM kernel via in-kernel cycle stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'avt_amd', 'libavt_hip_lab.so'))   # lab build: make -C avt_amd/csrc lab
dbg = torch.zeros(4 * 20000, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
M = 63040
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
for name, N, K, kw in [('proj', 768, 768, {}), ('proj+bias+res', 768, 768, 'res'), ('fc1+bias+gelu+c2', 3072, 768, 'gelu'), ('fc1+bias+gelu', 3072, 768, 'gelu1'), ('fc1+bias+c2', 3072, 768, 'c2only'), ('fc1 plain', 3072, 768, {}), ('fc2dgrad aux', 3072, 768, 'aux'), ('fc2dgrad aux+colsum', 3072, 768, 'auxcs'), ('fc2', 768, 3072, {})]:
    x, w = r((M, K)), r((N, K))
    for tile in (256,):
        kws = {}
        if kw == 'gelu':
            kws = dict(bias=torch.zeros(N, device='cuda'), act=1, c2=torch.empty((M, N), device='cuda', dtype=torch.bfloat16))
        if kw == 'gelu1':
            kws = dict(bias=torch.zeros(N, device='cuda'), act=1)
        if kw == 'c2only':
            kws = dict(bias=torch.zeros(N, device='cuda'), c2=torch.empty((M, N), device='cuda', dtype=torch.bfloat16))
        if kw == 'res':
            kws = dict(bias=torch.zeros(N, device='cuda'), res=r((M, N)))
        if kw in ('aux', 'auxcs'):
            kws = dict(act=3, aux=r((M, N)))
            if kw == 'auxcs': kws['colsum'] = torch.zeros(N, device='cuda')
        for _ in range(2):
            dbg.zero_()
            ops.gemm(x, w, M, N, K, tile=tile, **kws)
            torch.cuda.synchronize()
        tl = 256 if tile == 808 else tile
        nb = ((M + tl - 1) // tl) * ((N + tl - 1) // tl)
        d = dbg[:4 * nb].view(nb, 4).double()
        loop = (d[:, 1] - d[:, 0]); epi = (d[:, 2] - d[:, 1]); span = (d[:, 2].max() - d[:, 0].min())
        print(f'{name:18s} tile={tile}: blocks={nb} nk={int(d[0,3])} loop avg {loop.mean():8.0f} ticks  epilogue avg {epi.mean():8.0f} ticks  epi/(loop+epi)={float(epi.mean()/(loop.mean()+epi.mean())):.2f}  kernel span {span:9.0f} ticks')
