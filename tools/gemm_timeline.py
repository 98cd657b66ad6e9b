"""Lab: per-block prologue+loop / epilogue timing of the generic GEMM kernel via in-kernel cycle stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dbg = torch.zeros(4 * 20000, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
M = 63040
r = lambda s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
for name, N, K, kw in [('proj', 768, 768, {}), ('fc1+bias+gelu+c2', 3072, 768, 'gelu'), ('fc1 plain', 3072, 768, {}), ('fc2', 768, 3072, {})]:
    x, w = r((M, K)), r((N, K))
    for tile in (128, 256):
        kws = {}
        if kw == 'gelu':
            kws = dict(bias=torch.zeros(N, device='cuda'), act=1, c2=torch.empty((M, N), device='cuda', dtype=torch.bfloat16))
        for _ in range(2):
            dbg.zero_()
            ops.gemm(x, w, M, N, K, tile=tile, **kws)
            torch.cuda.synchronize()
        nb = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
        d = dbg[:4 * nb].view(nb, 4).double()
        loop = (d[:, 1] - d[:, 0]); epi = (d[:, 2] - d[:, 1]); span = (d[:, 2].max() - d[:, 0].min())
        print(f'{name:18s} tile={tile}: blocks={nb} nk={int(d[0,3])} loop avg {loop.mean():8.0f} ticks  epilogue avg {epi.mean():8.0f} ticks  epi/(loop+epi)={float(epi.mean()/(loop.mean()+epi.mean())):.2f}  kernel span {span:9.0f} ticks')
