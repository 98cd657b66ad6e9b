"""lab: is a hipGraph replay of the whole training step faster than eager launches at small batch?  (seeds / LR baked in:
a feasibility probe only, not a training mode).  usage: python tools/lab/graph_probe.py BATCH"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = types.SimpleNamespace(model='vit_base_patch16_224', batch=B, frames=10, bucket_mb=64, reduce_mode='all_reduce', wire_dtype='fp32')
dev = torch.device('cuda', 0)
trainer, data = bench.build(args, dev, 1)
for _ in range(4):
    trainer.step(data)
torch.cuda.synchronize()
def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = timed(lambda: trainer.step(data))
print(f'B={B}: eager {eager:.2f} ms/step = {B / eager * 1e3:.1f} clips/s', flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        trainer.step(data)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    loss, _, _, _ = trainer.step(data)
torch.cuda.synchronize()
print(f'captured in {time.perf_counter() - t0:.2f} s', flush=True)
rep = timed(g.replay)
print(f'B={B}: graph replay {rep:.2f} ms/step = {B / rep * 1e3:.1f} clips/s  (loss {float(loss):.4f})', flush=True)
