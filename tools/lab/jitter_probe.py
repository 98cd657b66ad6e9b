"""Lab: which Pillow operation differs?  Identity geometry, one operation (or a pair) at a time, device chain vs oracle."""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import ops
from oracle import avt_oracle as O
g = torch.Generator().manual_seed(3)
T, H, W = 3, 72, 128
clip = torch.randint(0, 256, (1, T, H, W, 3), generator=g, dtype=torch.uint8)
names = ('brightness', 'contrast', 'saturation', 'hue')
def run(chain, flip=0):
    ids = [names.index(n) for n, _ in chain] + [-1] * (4 - len(chain))
    fs = [float(int(f * 255) & 255) if n == 'hue' else f for n, f in chain] + [0.] * (4 - len(chain))
    params = torch.tensor([[H, W, flip, 0, 0, 0]], dtype=torch.int32).cuda()
    out = ops.video_preproc_jitter(clip.cuda(), params, torch.tensor([ids], dtype=torch.int32).cuda(), torch.tensor([fs], dtype=torch.float32).cuda(),
                                   (H, W), mean=(0, 0, 0), std=(1, 1, 1))
    dev = (out[0, :, :, 0] * 255).round().cpu()                      # (T, 3, H, W) levels
    ref = O.video_preproc(clip[0], (H, W), flip, (0, 0), (H, W), mean=(0, 0, 0), std=(1, 1, 1), color_jitter_ops=list(chain))
    ref = (ref.permute(1, 0, 2, 3) * 255).round()
    d = (dev - ref).abs()
    print(chain, 'flip', flip, 'max', float(d.max()), 'frac', float((d > 0).float().mean()))
for c in [[('brightness', 0.77)], [('contrast', 1.27)], [('saturation', 1.33)], [('hue', 0.1)], [('hue', -0.07)],
          [('saturation', 1.33), ('hue', 0.1)], [('hue', 0.1), ('brightness', 0.77)], [('brightness', 0.77), ('contrast', 1.27)],
          [('saturation', 1.33), ('hue', 0.1), ('brightness', 0.77), ('contrast', 1.27)]]:
    run(c)
run([('contrast', 1.27)], flip=1)
