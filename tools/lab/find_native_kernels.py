"""Which host-side torch calls still launch their own kernels / copies inside a training step?  Runs two steps of the bench configuration under
torch.profiler and prints, for every device kernel or memcpy that is not one of this library's, the Python stack of the op that launched it.
usage: python tools/lab/find_native_kernels.py [batch]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse, torch
import bench

a = argparse.Namespace(batch=int(sys.argv[1]) if len(sys.argv) > 1 else 32, frames=10, model='vit_base_patch16_224', bucket_mb=64, reduce_mode='all_reduce',
                       wire_dtype='fp32', tail_mb=-1)
dev = torch.device('cuda', 0)
trainer, data = bench.build(a, dev, 1)
for _ in range(3):
    trainer.step(data)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        trainer.step(data)
    torch.cuda.synchronize()
ours = ('gemm_', 'vit_attn', 'cls_attn', 'causal_attn', 'ln_', 'sgd_kernel', 'splitk_reduce', 'partials_reduce', 'im2col', 'dropout_kernel', 'embed_pos', 'patch_bwd', 'colsum_kernel',
        'cast_kernel', 'transpose', 'xent', 'posres', 'mse_shift', 'pad_cast', 'add_rows', 'relu_kernel', 'video_preproc', 'avt_')
ev = prof.events()
by_op = collections.defaultdict(lambda: [0, 0.0, None])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
    for k in e.kernels:
        if any(o in k.name for o in ours):
            continue
        key = (e.name, k.name[:70], tuple(str(s) for s in (e.stack or [])[:6]), str(e.input_shapes)[:80])
        d = by_op[key]
        d[0] += 1; d[1] += k.duration
for (op, kern, stack, shapes), (n, us, _) in sorted(by_op.items(), key=lambda kv: -kv[1][1]):
    print(f'{n / 2:6.1f}/step {us / 2:9.1f} us/step  {op:28s} {kern}\n        shapes {shapes}')
    for s in stack:
        if 'site-packages' not in s:
            print('        ', s)
