"""Where a wave of the single-pass attention backward spends an item (lab library only: AVT_HIP_LIB=.../libavt_hip_lab.so).
The kernel stamps s_memtime before / after each of an item's 8 barriers and after each dQ product (vit_attention.hip, AVT_BWD1_STAMP), for
the 4th item of every workgroup.  Printed: per chunk, averaged over the workgroups,
   work   = arrival at barrier c  - (release of barrier c-1, or the end of the wave's dQ product of chunk c-1)
   wait   = release - arrival                     (split into waves that had a dQ product in the previous chunk and the others)
   dq     = end of the dQ product - release       (the 4 waves that have one)
and the item's critical path.  usage: attn_timeline.py [frames] [scaled 0|1]"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from avt_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
scaled = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S, H, D = 197, 12, 768
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(5)
qkv = (torch.randn(frames * S, 3 * D, device=dev, generator=g) * 0.7).bfloat16()
dout = (torch.randn(frames * S, D, device=dev, generator=g) * 0.05).bfloat16()
out, lse = ops.vit_attn_fwd(qkv, frames, S, H)
dbias = torch.zeros(3 * D, device=dev)
stat = torch.rand(frames * S, 2, device=dev, generator=g) + 0.5 if scaled else None


def run():
    return ops.vit_attn_bwd(qkv, out, dout, lse, frames, S, H, dbias=dbias, row_stat=stat)


ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3): ops.vit_attn_fwd(qkv, frames, S, H)
ev[0].record()
for _ in range(10): ops.vit_attn_fwd(qkv, frames, S, H)
ev[1].record(); torch.cuda.synchronize()
print(f'frames {frames}: forward {ev[0].elapsed_time(ev[1]) * 100:.1f} us per launch')
for _ in range(3): run()
torch.cuda.synchronize()
ev[0].record()
for _ in range(10): run()
ev[1].record(); torch.cuda.synchronize()
print(f'frames {frames} scaled {scaled}: {ev[0].elapsed_time(ev[1]) * 100:.1f} us per launch (stamps off; includes the bias partial reduce)')

buf = torch.zeros(256 * 16 * 32, device=dev, dtype=torch.int32)
os.environ['AVT_BWD1_STAMPS_PTR'] = str(buf.data_ptr())
ev[0].record(); run(); ev[1].record(); torch.cuda.synchronize()
print(f'with stamps: {ev[0].elapsed_time(ev[1]) * 1000:.1f} us')
os.environ['AVT_BWD1_STAMPS_PTR'] = '0'
run(); torch.cuda.synchronize()
if not buf.any().item():
    print('no stamps (not the lab library): done'); sys.exit(0)
st = buf.cpu().numpy().astype(np.int64).reshape(256, 16, 32)[:, :13, :24] & 0xffffffff
NP = 7
ok = st[:, :, 0].min(axis=1) > 0
print(f'workgroups with stamps: {int(ok.sum())} of 256')
st = st[ok]
t0 = st[:, :, 0].min(axis=1)[:, None, None]
rel = (st - t0) & 0xffffffff                     # cycles (s_memtime: 100 MHz-independent shader clock counter) since the first wave reached barrier S
A = lambda c: rel[:, :, 2 + 3 * c]
R = lambda c: rel[:, :, 3 + 3 * c]
Q = lambda c: rel[:, :, 4 + 3 * c]
print('item = barrier S .. end: %.0f cycles (mean over workgroups of the last wave\'s end stamp)' % rel[:, :, 23].max(axis=1).mean())
print('barrier S: arrival spread %.0f, release %.0f' % ((rel[:, :, 0].max(axis=1) - rel[:, :, 0].min(axis=1)).mean(), rel[:, :, 1].mean()))
prev_end = np.broadcast_to(rel[:, :, 1], rel[:, :, 1].shape).copy()
waves = np.arange(13)
print('chunk |  work(all)  work(dq waves of c-1: their dq excluded) |  dq product |  wait(dq waves of c-1)  wait(others) | period (release c - release c-1)')
prev_rel = rel[:, :, 1]
for c in range(NP):
    dqw_prev = np.zeros(13, bool)
    if c > 0:
        for h in range(4 * (c - 1), 4 * (c - 1) + 4):
            if h < 26: dqw_prev[h % 13] = True
    dqw = np.zeros(13, bool)
    for h in range(4 * c, 4 * c + 4):
        if h < 26: dqw[h % 13] = True
    work = A(c) - prev_end
    wait = R(c) - A(c)
    dq = Q(c) - R(c)
    period = (R(c) - prev_rel).mean()
    w_all = work.mean()
    w_dq = work[:, dqw_prev].mean() if dqw_prev.any() else float('nan')
    wt_dq = wait[:, dqw_prev].mean() if dqw_prev.any() else float('nan')
    wt_ot = wait[:, ~dqw_prev].mean()
    last = A(c).argmax(axis=1)                   # which wave arrives last
    frac_last_dq = dqw_prev[last].mean() if dqw_prev.any() else float('nan')
    print(f'  {c}   | {w_all:8.0f} {w_dq:8.0f} | {dq[:, dqw].mean():8.0f} (others {dq[:, ~dqw].mean():5.0f}) | {wt_dq:8.0f} {wt_ot:8.0f} | {period:8.0f}   last arriver is a dq wave of c-1: {frac_last_dq:.2f}')
    prev_end = Q(c).copy()
    prev_rel = R(c)
print('tail (end stamp - last dq / release): %.0f' % (rel[:, :, 23] - prev_end).mean())
b = 0
print('workgroup 0, per wave: A_c - R_(c-1)/Q_(c-1) [work], R_c - A_c [wait], Q_c - R_c [dq]')
for w in range(13):
    row = []
    pe = rel[b, w, 1]
    for c in range(NP):
        row.append(f'{rel[b, w, 2 + 3 * c] - pe:5d}/{rel[b, w, 3 + 3 * c] - rel[b, w, 2 + 3 * c]:5d}/{rel[b, w, 4 + 3 * c] - rel[b, w, 3 + 3 * c]:5d}')
        pe = rel[b, w, 4 + 3 * c]
    print(f'  w{w:2d}: ' + '  '.join(row))
