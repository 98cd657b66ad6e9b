"""Lab: per-workgroup timeline of a GEMM launch from in-kernel cycle stamps (libavt_hip_lab.so, AVT_GEMM_DBG_PTR): start / end of
the K loop / epilogue arithmetic issued / stores drained, plus the placement (XCC, SE, CU, wave slot).  Answers: how long is a
tile's K loop, its epilogue and its store drain at the bench's shapes; does a CU start its next tile immediately; are the CUs of
the chip in the same phase at the same time (all in the epilogue together = synchronized store bursts while the matrix pipes idle)?
usage: python tools/gemm_timeline.py [tile ...]   (808 = 8-phase, 2562 / 2563 = two 4-wave workgroups per CU; AVT_GEMM_STAGGER=N)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault('AVT_HIP_LIB', os.path.join(ROOT, 'avt_amd', 'libavt_hip_lab.so'))
NB = 100000
dbg = torch.zeros(16 * NB, device='cuda', dtype=torch.int64)
os.environ['AVT_GEMM_DBG_PTR'] = hex(dbg.data_ptr())
from avt_amd import ops
B = int(os.environ.get('KB_BATCH', 256)); M = B * 10 * 197
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x, w1 = r(M, 768), r(3072, 768)
b3 = torch.rand(3072, device='cuda')
pre = torch.empty((M, 3072), device='cuda', dtype=torch.bfloat16); act = torch.empty_like(pre)
cases = {'fc1 gelu+c2': dict(bias=b3, act=ops.ACT_GELU_ERF, c2=pre), 'fc1 plain': dict()}
if os.environ.get('TL_CASES'):
    cases = {k: v for k, v in cases.items() if k in os.environ['TL_CASES'].split(',')}
for tile in [int(t) for t in sys.argv[1:]] or [808]:
    for name, kw in cases.items():
        for _ in range(2):
            dbg.zero_()
            ops.gemm(x, w1, M, 3072, 768, tile=tile, out=act, **kw)
            torch.cuda.synchronize()
        d = dbg.view(-1, 8)
        d = d[d[:, 0] > 0].cpu().double()
        # every XCD has its own counter base: cluster the records by base (sorted start times, split at jumps > 1e7 ticks) and
        # measure time from each cluster's first start (all XCDs start a launch within microseconds of each other)
        order = torch.argsort(d[:, 0]); d = d[order]
        jump = torch.cat([torch.zeros(1), (d[1:, 0] - d[:-1, 0] > 1e7).double()]).cumsum(0).long()      # cluster index = XCD
        base = torch.zeros(int(jump.max()) + 1, dtype=torch.float64).scatter_reduce(0, jump, d[:, 0], 'amin', include_self=False)
        start, loop, math_, end = (d[:, i] - base[jump] for i in range(4))
        hw = d[:, 4].long(); xraw = d[:, 5].long()
        cu = (jump << 8) | ((hw >> 8) & 0x7f)                  # XCD cluster | SE / SH / CU bits of HW_ID
        slot = hw & 15; simd = (hw >> 4) & 3; tg = (hw >> 16) & 15
        span = float(end.max())
        kl, ep, dr = (loop - start), (math_ - loop), (end - math_)
        print(f'== tile {tile} {name} stagger {os.environ.get("AVT_GEMM_STAGGER", "0")}: {len(d)} records in {int(jump.max()) + 1} counter clusters (XCDs), kernel span {span:.0f} ticks; per record: K loop {kl.mean():.0f} (sd {kl.std():.0f})  epilogue issue {ep.mean():.0f} (sd {ep.std():.0f})  store drain {dr.mean():.0f}')
        print(f'   wave slots {sorted(set(slot.tolist()))}  SIMDs {sorted(set(simd.tolist()))}  TG ids {sorted(set(tg.tolist()))[:8]}  raw XCC_ID values {sorted(set(xraw.tolist()))[:10]}  distinct CUs {len(set(cu.tolist()))}')
        import collections
        by = collections.defaultdict(list)
        for i in range(len(d)):
            by[int(cu[i])].append((float(start[i]), float(loop[i]), float(end[i]), int(slot[i]), int(tg[i])))
        # per CU: how much of the time is at least one record in its K loop, and how much of every epilogue runs while another
        # record on the same CU is in its K loop (two workgroups per CU: the point of the exercise)
        cov_loop, epi_overlap, gaps = [], [], []
        for k, v in by.items():
            v.sort()
            t_lo, t_hi = v[0][0], max(x[2] for x in v)
            ev = sorted([(a, 1) for a, b, c, _, _ in v] + [(b, -1) for a, b, c, _, _ in v])
            busy, depth, last = 0.0, 0, t_lo
            for t, dlt in ev:
                if depth > 0: busy += t - last
                depth += dlt; last = t
            cov_loop.append(busy / (t_hi - t_lo))
            for a, b, c, sl_, _ in v:
                ov = sum(max(0.0, min(c, b2) - max(b, a2)) for a2, b2, c2, sl2, _ in v if (a2, b2) != (a, b) and b2 > b and a2 < c)
                epi_overlap.append(ov / max(c - b, 1.0))
            per_slot = collections.defaultdict(list)
            for a, b, c, sl_, tg_ in v: per_slot[(sl_, tg_)].append((a, c))
            for vv in per_slot.values():
                gaps += [y[0] - x[1] for x, y in zip(vv[:-1], vv[1:])]
        print(f'   per CU: some workgroup is inside its K loop {100 * sum(cov_loop) / len(cov_loop):.1f} % of the launch; fraction of an epilogue that runs under ANOTHER record\'s K loop on the same CU: mean {sum(epi_overlap) / len(epi_overlap):.3f}')
        g = torch.tensor(gaps)
        print(f'   gap between a record\'s drained stores and the next start in the same (CU, slot): median {g.median():.0f}  p90 {g.quantile(0.9):.0f}')
        ts = torch.linspace(0.2 * span, 0.8 * span, 400)
        in_loop = torch.stack([((start <= t) & (loop > t)).sum() for t in ts]).double()
        in_epi = torch.stack([((loop <= t) & (end > t)).sum() for t in ts]).double()
        fl = in_loop / (in_loop + in_epi).clamp(min=1)
        print(f'   census over the middle of the launch: fraction of resident records in the K loop mean {fl.mean():.3f}  min {fl.min():.3f}  max {fl.max():.3f}  sd {fl.std():.3f} (0 = evenly mixed phases, large = the chip breathes in lock-step)')
