import numpy as np, torch, torch.nn.functional as F
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
def idx(insz, outsz):
    scale = f32(insz) / f32(outsz)
    d = np.arange(outsz, dtype=np.float32) + f32(0.5)
    r = np.maximum(fma(np.full_like(d, scale), d, np.full_like(d, -0.5)), f32(0))
    i0 = np.minimum(np.floor(r).astype(np.int64), insz - 1)
    lam = np.clip(r - i0.astype(np.float32), f32(0), f32(1)).astype(np.float32)
    return i0, i0 + (i0 < insz - 1), (f32(1) - lam).astype(np.float32), lam
def resize(x, oh, ow):
    H, W = x.shape[-2:]
    y0, y1, h0, h1 = idx(H, oh); x0, x1, w0, w1 = idx(W, ow)
    p00, p01 = x[..., y0[:, None], x0[None, :]], x[..., y0[:, None], x1[None, :]]
    p10, p11 = x[..., y1[:, None], x0[None, :]], x[..., y1[:, None], x1[None, :]]
    bc = lambda v, ax: np.broadcast_to(v[None, :] if ax else v[:, None], p00.shape)
    top = fma(p00, bc(w0, 1), (p01 * bc(w1, 1)).astype(f32)); bot = fma(p10, bc(w0, 1), (p11 * bc(w1, 1)).astype(f32))
    return fma(top, bc(h0, 0), (bot * bc(h1, 0)).astype(f32))
g = torch.Generator().manual_seed(0)
tot = 0
for (H, W, oh, ow) in [(256, 456, 248, 441), (256, 456, 280, 498), (64, 96, 56, 84), (240, 320, 271, 361), (256, 456, 256, 456), (480, 854, 248, 441)]:
    u8 = torch.randint(0, 256, (3, 4, H, W), generator=g, dtype=torch.uint8)
    x = u8.float() / 255.0
    ref = F.interpolate(x, size=(oh, ow), mode='bilinear').numpy()
    mine = resize(x.numpy(), oh, ow)
    print((H, W, oh, ow), 'float mismatches', int((mine != ref).sum()), 'of', ref.size)
torch.set_num_threads(1)
ref1 = F.interpolate(x, size=(oh, ow), mode='bilinear').numpy()
print('1 thread same:', bool((ref1 == ref).all()))
