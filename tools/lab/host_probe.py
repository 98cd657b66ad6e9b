"""Lab: where does the host spend a training step -- is it blocked by the device somewhere (enqueue time per region, with the device idle
vs busy), and does the caching allocator retry / re-allocate per step?   usage: python tools/lab/host_probe.py [batch=256]"""
import sys, time, argparse
sys.path.insert(0, '.')
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = argparse.Namespace(model='vit_base_patch16_224', batch=B, frames=10, bucket_mb=64, reduce_mode='all_reduce', wire_dtype='fp32', tail_mb=-1)
dev = torch.device('cuda', 0)
trainer, data = bench.build(args, dev, 1)
for _ in range(3):
    trainer.step(data)
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats()
regions = {}
def lap(name, t0):
    regions[name] = regions.get(name, 0.0) + time.perf_counter() - t0
N = 5
tt = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); d, outputs, losses, acc = trainer.op(data, train_mode=True); lap('forward (op)', t0)
    t0 = time.perf_counter(); loss = trainer.total_loss(losses); trainer.optimizer.zero_grad(); lap('loss', t0)
    t0 = time.perf_counter(); loss.backward(); lap('backward', t0)
    t0 = time.perf_counter(); trainer.optimizer.grad_scale = 1.0; trainer.optimizer.step(); lap('optimizer', t0)
enq = time.perf_counter() - tt
torch.cuda.synchronize()
tot = time.perf_counter() - tt
s1 = torch.cuda.memory_stats()
print(f'{N} steps: host enqueue {enq / N * 1e3:.1f} ms/step, wall {tot / N * 1e3:.1f} ms/step')
for k, v in regions.items():
    print(f'   {k:16s} {v / N * 1e3:8.1f} ms/step')
for k in ('num_alloc_retries', 'num_ooms', 'num_device_alloc', 'num_device_free', 'allocation.all.allocated', 'segment.all.allocated'):
    print(f'   {k}: {s0.get(k)} -> {s1.get(k)}')
print(f'   reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, allocated peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
# the same with the device idle at every region boundary: pure host cost of enqueuing
regions.clear()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d, outputs, losses, acc = trainer.op(data, train_mode=True); lap('forward (op)', t0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); loss = trainer.total_loss(losses); lap('loss', t0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); loss.backward(); lap('backward', t0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); trainer.optimizer.step(); lap('optimizer', t0)
print('enqueue with an idle device at every region start (includes waiting wherever a call blocks):')
for k, v in regions.items():
    print(f'   {k:16s} {v / 2 * 1e3:8.1f} ms/step')
