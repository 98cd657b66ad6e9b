"""LayerNorm backward / forward per-launch time at the step's shape (504320 x 768) with the library given by AVT_HIP_LIB."""
import sys, torch
sys.path.insert(0, '.')
from avt_amd import ops
rows, D = 2560 * 197, 768
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(s, device='cuda', generator=g).to(torch.bfloat16)
x, dy, dres = r(rows, D), r(rows, D), r(rows, D)
gamma, beta = torch.rand(D, device='cuda') + 0.5, torch.rand(D, device='cuda')
y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
dg, db, cs = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
best = min(t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, colsum=cs)) for _ in range(3))
dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, colsum=cs)
print(f'ln_bwd {best:7.1f} us  {rows * D * 8 / best / 1e6:5.2f} TB/s   checksum {float(dx.float().abs().sum()):.6e} {float(dg.sum()):.6e}', flush=True)
bf = min(t(lambda: ops.layernorm_fwd(x, gamma, beta, 1e-6)) for _ in range(3))
print(f'ln_fwd {bf:7.1f} us  {rows * D * 4 / bf / 1e6:5.2f} TB/s', flush=True)
