"""Lab: per-tile timeline of the persistent 8-phase GEMM from in-kernel cycle stamps (libavt_hip_lab.so, AVT_GEMM_DBG_PTR): K loop,
epilogue issue, wait + barrier before the next tile, and the phase-2 wait of a tile's first iteration (= the previous tile's store drain).
usage: python tools/lab/persist_timeline.py [frames=2560]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(ROOT, 'avt_amd', 'libavt_hip_lab.so'))
dbg = torch.zeros(256 * 128 * 2 * 8 + 256 * 16, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
M = frames * 197; M -= M % 256
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x768, x3072 = r(M, 768), r(M, 3072)
resid, pre = r(M, 768), r(M, 3072)
cases = [('qkv fwd (bias)', x768, 2304, dict(bias=torch.rand(2304, device='cuda'))),
         ('fc1 dgrad (plain, K=3072)', x3072, 768, dict()),
         ('fc1 fwd (gelu + c2)', x768, 3072, dict(bias=torch.rand(3072, device='cuda'), act=ops.ACT_GELU_ERF, c2=torch.empty((M, 3072), device='cuda', dtype=torch.bfloat16))),
         ('proj fwd (bias + res)', x768, 768, dict(bias=torch.rand(768, device='cuda'), res=resid)),
         ('fc2 dgrad (* aux, colsum)', x768, 3072, dict(act=ops.ACT_MUL_AUX, aux=pre, colsum=torch.zeros(3072, device='cuda')))]
for name, A, N, kw in cases:
    K = A.size(1)
    W = r(N, K)
    for _ in range(2):
        dbg.zero_()
        ops.linear_fwd(A, W, **kw)
        torch.cuda.synchronize()
    itr = dbg[256 * 128 * 2 * 8:].view(256, 16).cpu().double()
    raw = dbg[:256 * 128 * 2 * 8].view(256, 128, 2, 8).cpu().double()
    d = raw.view(-1, 8)
    d = d[d[:, 0] > 0]
    top, loop, epi, nxt, w2, k = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 7]
    mid = (k > 0) & (k < k.max())
    f = lambda v: f'{v[mid].mean():8.0f} (sd {v[mid].std():6.0f})'
    # tile period = start of tile k+1 - start of tile k of the same workgroup and wave group
    t0 = raw[:, :-1, :, 0]; t1 = raw[:, 1:, :, 0]
    ok = (t0 > 0) & (t1 > 0)
    per = (t1 - t0)[ok]
    gap = (raw[:, 1:, :, 0] - raw[:, :-1, :, 3])[ok]
    span = (d[:, 3].max() - d[:, 0].min())
    print(f'== {name}: {len(d)} records, tiles per workgroup up to {int(k.max()) + 1}; cycles:')
    print(f'   K loop {f(loop - top)}   epilogue issue {f(epi - loop)}   wait+barrier {f(nxt - epi)}   tile total {f(nxt - top)}   phase-2 wait of iteration 0 {f(w2)}')
    ok_i = itr[:, 0] > 0
    if ok_i.any() and K == 768:
        it = itr[ok_i]
        seq = torch.cat([it[:, 9:10], it[:, 0:6], it[:, 8:9]], 1)          # tile top, iteration starts 0..5, end of loop
        dd = (seq[:, 1:] - seq[:, :-1]).mean(0)
        print('   4th tile of each workgroup: top -> it0 ' + ' '.join(f'{v:6.0f}' for v in dd.tolist()) + '   (iteration 0..5 durations; the last = it5 + end-of-loop barriers)')
    print(f'   tile period {per.mean():8.0f} (sd {per.std():6.0f})   between tiles {gap.mean():6.0f} (sd {gap.std():5.0f})', flush=True)
