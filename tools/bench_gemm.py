"""Micro-benchmark of the GEMM shapes of one ViT-B layer + head (uniform random operands). Prints TF/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avt_amd import ops

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def r(shape): return (torch.rand(shape, device='cuda') * 2 - 1).to(torch.bfloat16)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M = B * 10 * 197
print(f'M = {M}')
for name, N, K in [('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    x, w, dy = r((M, K)), r((N, K)), r((M, N))
    dw = torch.zeros((N, K), device='cuda')
    bias = torch.zeros(N, device='cuda')
    fl = 2.0 * M * N * K
    for tile in (256, 808):
        t = bench(lambda: ops.linear_fwd(x, w, bias=bias, tile=tile)); print(f'{name:5s} fwd   NT tile={tile} {fl/t/1e12:7.1f} TF/s  {t*1e6:8.1f} us')
        t = bench(lambda: ops.linear_dgrad(dy, w, tile=tile));          print(f'{name:5s} dgrad NN tile={tile} {fl/t/1e12:7.1f} TF/s  {t*1e6:8.1f} us')
    for tile, sk in ((256, 0), (808, 0)):
        t = bench(lambda: ops.linear_wgrad(dy, x, dw, splitk=sk, tile=tile));   print(f'{name:5s} wgrad TN tile={tile} splitk={sk:2d} {fl/t/1e12:7.1f} TF/s  {t*1e6:8.1f} us')
Mh = B * 10
for name, N, K in [('c_attn', 6144, 2048), ('c_proj', 2048, 2048), ('c_fc', 8192, 2048), ('mlp_proj', 2048, 8192)]:
    x, w, dy = r((Mh, K)), r((K, N)), r((Mh, N))
    dw = torch.zeros((K, N), device='cuda')
    by = (K * N * 2.0)
    for tile in (64, 643, 128):
        t = bench(lambda: ops.conv1d_fwd(x, w, tile=tile)); print(f'{name:8s} fwd   tile={tile} {t*1e6:8.1f} us  {by/t/1e12:5.2f} TB/s(weights)')
        t = bench(lambda: ops.conv1d_dgrad(dy, w, tile=tile)); print(f'{name:8s} dgrad tile={tile} {t*1e6:8.1f} us  {by/t/1e12:5.2f} TB/s')
    t = bench(lambda: ops.conv1d_wgrad(x, dy, dw)); print(f'{name:8s} wgrad {t*1e6:8.1f} us  {2*by/t/1e12:5.2f} TB/s(fp32 out)')
# other kernels
D = 768
x = r((M, D)); g = torch.ones(D, device='cuda'); b = torch.zeros(D, device='cuda')
t = bench(lambda: ops.layernorm_fwd(x, g, b, 1e-6)); print(f'ln fwd  {t*1e6:8.1f} us {M*D*4/t/1e12:5.2f} TB/s')
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
dg, db, cs = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
t = bench(lambda: ops.layernorm_bwd(x, x, mean, rstd, g, dg, db, dres=x, colsum=cs)); print(f'ln bwd  {t*1e6:8.1f} us {M*D*8/t/1e12:5.2f} TB/s')
qkv = r((M, 3 * D)); frames = B * 10
t = bench(lambda: ops.vit_attn_fwd(qkv, frames, 197, 12)); fl = 4.0 * 197 * 197 * 64 * 12 * frames
print(f'attn fwd {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s')
o, lse = ops.vit_attn_fwd(qkv, frames, 197, 12)
dbias = torch.zeros(3 * D, device='cuda')
t = bench(lambda: ops.vit_attn_bwd(qkv, o, o, lse, frames, 197, 12, dbias=dbias)); print(f'attn bwd {t*1e6:8.1f} us {2.5*fl/t/1e12:6.1f} TF/s')
v = torch.rand((frames, 3, 224, 224), device='cuda')
t = bench(lambda: ops.im2col_patch16(v)); print(f'im2col {t*1e6:8.1f} us')
n = 396_000_000
p_, g_, m_ = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
sh = torch.empty(n, device='cuda', dtype=torch.bfloat16)
t = bench(lambda: ops.sgd_step(p_, g_, m_, sh, 0.1, 0.9, 1e-6), iters=5); print(f'sgd 396M {t*1e6:8.1f} us {n*26/t/1e12:5.2f} TB/s')
