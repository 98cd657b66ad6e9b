"""Where a wave of the single-pass attention backward spends an item (lab library only: AVT_HIP_LIB=.../libavt_hip_lab.so).
The kernel stamps s_memtime before / after each of an item's 8 barriers and after each dQ product (vit_attention.hip, AVT_BWD1_STAMP), for
the 4th item of every workgroup.  Printed: per chunk, averaged over the workgroups,
   work   = arrival at barrier c  - (release of barrier c-1, or the end of the wave's dQ product of chunk c-1)
   wait   = release - arrival                     (split into waves that had a dQ product in the previous chunk and the others)
   dq     = end of the dQ product - release       (the 4 waves that have one)
and the item's critical path.  usage: attn_timeline.py [frames] [scaled 0|1] [heads]   (heads = 1 with 12 x the frames: the same items on rows of 384 contiguous bytes)"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from avt_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
scaled = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
S, D = 197, H * 64
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
st = buf.cpu().numpy().astype(np.int64).reshape(256, 16, 32)[:, :13, :] & 0xffffffff
ok = st[:, :, 0].min(axis=1) > 0
print(f'workgroups with stamps: {int(ok.sum())} of 256')
st = st[ok]
t0 = st[:, :, 0].min(axis=1)[:, None, None]
rel = (st - t0) & 0xffffffff                     # cycles since the first wave reached barrier S
have = lambda i: bool((st[:, :, i] != 0).any())
m = lambda a: float(np.mean(a))
print('stamps present:', [i for i in range(32) if have(i)])
print(f'barrier S : arrival spread {m(rel[:, :, 0].max(axis=1)):.0f}, released {m(rel[:, :, 1]):.0f} after the first arrival')
if have(24):
    print(f'D of the strips from LDS: {m(rel[:, :, 24] - rel[:, :, 1]):.0f};  barrier S2 wait {m(rel[:, :, 25] - rel[:, :, 24]):.0f};  S2 released at {m(rel[:, :, 25]):.0f}')
start = rel[:, :, 25] if have(24) else rel[:, :, 1]
print(f'chunk 0   : work {m(rel[:, :, 2] - start):.0f}, wait {m(rel[:, :, 3] - rel[:, :, 2]):.0f};  barrier 0 released at {m(rel[:, :, 3]):.0f}')
if have(20):
    print(f'chunks 1-6: barrier 0 release -> barrier 6 arrival {m(rel[:, :, 20] - rel[:, :, 3]):.0f} (wait at 6: {m(rel[:, :, 21] - rel[:, :, 20]):.0f});  barrier 6 released at {m(rel[:, :, 21]):.0f}')
    dq = rel[:, :, 22] - rel[:, :, 21]
    print(f'after barrier 6: dQ products of waves 11 / 12 {m(dq[:, 11]):.0f} / {m(dq[:, 12]):.0f} (others {m(dq[:, :11]):.0f});  tail (-> stores issued) {m(rel[:, :, 23] - rel[:, :, 22]):.0f}')
    print(f'end stamp: first wave {m(rel[:, :, 23].min(axis=1)):.0f}, last wave {m(rel[:, :, 23].max(axis=1)):.0f};  per wave w0..w12: ' + ' '.join(f'{m(rel[:, w, 23]):.0f}' for w in range(13)))
print('(the item period is the launch time / items per workgroup: %.0f items)' % (frames * H / 256))
