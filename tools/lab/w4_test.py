"""lab: the 4-wave weight-gradient kernel (tile 2565) against the 8-phase one on the step's four weight-gradient shapes (+ ragged cases):
results (fp32 accumulate, split boundaries differ: equal to ~1e-6 relative) and microseconds per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops
B = int(os.environ.get('KB_BATCH', 256)); M = B * 10 * 197
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
def run(name, rows, n_out, k_in, time_it=True):
    dy, x = r(rows, n_out), r(rows, k_in)
    outs, ts = [], []
    for tile in (0, 2565):
        ops.WGRAD_TILE = tile
        dw = torch.zeros((n_out, k_in), device='cuda')
        ops.linear_wgrad(dy, x, dw)
        torch.cuda.synchronize()
        outs.append(dw.clone())
        ts.append(timeit(lambda: ops.linear_wgrad(dy, x, dw)) if time_it else 0.0)
    ops.WGRAD_TILE = 0
    ref = None
    if rows <= 70000:
        ref = dy.float().t() @ x.float()
    e = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    er = '' if ref is None else f'  vs fp32 torch: 8p {float((outs[0]-ref).abs().max()/ref.abs().max()):.2e}  w4 {float((outs[1]-ref).abs().max()/ref.abs().max()):.2e}'
    fl = 2.0 * rows * n_out * k_in
    print(f'{name:34s} rows {rows:7d} -> [{n_out:5d} x {k_in:5d}]: 8p {ts[0]:8.1f} us ({fl/ts[0]/1e6 if ts[0] else 0:6.0f} TF/s)   w4 {ts[1]:8.1f} us ({fl/ts[1]/1e6 if ts[1] else 0:6.0f} TF/s)   max rel diff {e:.2e}{er}', flush=True)
run('ragged: rows % 32 != 0', 197 * 300 - 3, 768, 768, False)
run('ragged: out 2304 x in 776', 197 * 200 + 5, 2304, 776, False)
run('small: 1 K stage', 31, 256, 256, False)
run('fc1 wgrad', M, 3072, 768)
run('fc2 wgrad', M, 768, 3072)
run('qkv wgrad', M, 2304, 768)
run('proj wgrad', M, 768, 768)
