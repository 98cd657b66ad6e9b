"""Per-kernel timing on the bench shapes (HIP events, many launches): python tools/kbench.py [names...]
names: ln attn cls gemm wgrad sgd preproc (default: all of these); blas / sdpa time the vendor libraries on the same shapes.  Prints microseconds per launch and the achieved TB/s or TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avt_amd import ops

B, T, S, D, H = int(os.environ.get('KB_BATCH', 128)), 10, int(os.environ.get('KB_S', 197)), 768, 12
N = B * T
M = N * S
ops.DETERMINISTIC_WGRAD = os.environ.get('KB_DET', '1') == '1'
want = set(sys.argv[1:]) or {'ln', 'attn', 'cls', 'gemm', 'wgrad', 'sgd', 'preproc'}      # + 'blas': library GEMMs next to this repo's
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)


def timeit(name, fn, bytes_=None, flops=None, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    extra = ''
    if bytes_:
        extra += f'  {bytes_ / us / 1e6:6.2f} TB/s'
    if flops:
        extra += f'  {flops / us / 1e6:7.1f} TF/s'
    print(f'{name:34s} {us:9.1f} us{extra}', flush=True)
    return us


if 'ln' in want:
    x, dy, dres = r(M, D), r(M, D), r(M, D)
    g, b = torch.rand(D, device='cuda') + 0.5, torch.rand(D, device='cuda')
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    dg, db, cs = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    timeit('ln_fwd  M x 768', lambda: ops.layernorm_fwd(x, g, b, 1e-6), bytes_=M * D * 4)
    timeit('ln_bwd  M x 768 (+dres, colsum)', lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres, colsum=cs), bytes_=M * D * 8)
    del x, dy, dres, y
if 'attn' in want:
    qkv = r(M, 3 * D)
    out, lse = ops.vit_attn_fwd(qkv, N, S, H)
    do = r(M, D)
    dbias = torch.zeros(3 * D, device='cuda')
    fl = 4.0 * S * S * 64 * H * N
    timeit('vit_attn_fwd', lambda: ops.vit_attn_fwd(qkv, N, S, H), flops=fl)
    timeit('vit_attn_bwd', lambda: ops.vit_attn_bwd(qkv, out, do, lse, N, S, H, dbias=dbias), flops=2.5 * fl)
    del qkv, out, do
if 'cls' in want:
    q, kv, do = r(N, D), r(M, 2 * D), r(N, D)
    o, pr = ops.cls_attn_fwd(q, kv, N, S, H)
    timeit('cls_attn_fwd', lambda: ops.cls_attn_fwd(q, kv, N, S, H), bytes_=M * 2 * D * 2)
    timeit('cls_attn_bwd', lambda: ops.cls_attn_bwd(q, kv, pr, do, N, S, H), bytes_=M * 2 * D * 4)
    del q, kv, do
if 'gemm' in want:
  for TILE in [int(t) for t in os.environ.get('KB_TILES', '0').split(',')]:
    print(f'--- tile {TILE}'); ops.FORCE_TILE = TILE
    bias3, bias1 = torch.rand(3072, device='cuda'), torch.rand(768, device='cuda')
    x, w1, w2 = r(M, D), r(3072, D), r(D, 3072)
    pre = torch.empty((M, 3072), device='cuda', dtype=torch.bfloat16)
    act = torch.empty((M, 3072), device='cuda', dtype=torch.bfloat16)
    cs3 = torch.zeros(3072, device='cuda')
    fl = 2.0 * M * D * 3072
    timeit('fc1 fwd +bias+gelu+gelu\'', lambda: ops.linear_fwd(x, w1, bias=bias3, act=ops.ACT_GELU_ERF, c2=pre, out=act), flops=fl)
    timeit('fc1 fwd plain', lambda: ops.linear_fwd(x, w1, out=act), flops=fl)
    if 'epi' in want:
        timeit('  fc2 dgrad plain', lambda: ops.linear_dgrad(x, w2, out=act), flops=fl)
        timeit('  fc2 dgrad x aux', lambda: ops.linear_dgrad(x, w2, act=ops.ACT_MUL_AUX, aux=pre, out=act), flops=fl)
        timeit('  fc2 dgrad +colsum', lambda: ops.linear_dgrad(x, w2, colsum=cs3, out=act), flops=fl)
        timeit('  fc2 dgrad +res', lambda: ops.linear_dgrad(x, w2, res=pre, out=act), flops=fl)
        timeit('  fc1 fwd +bias', lambda: ops.linear_fwd(x, w1, bias=bias3, out=act), flops=fl)
        timeit('  fc1 fwd +bias+gelu (no c2)', lambda: ops.linear_fwd(x, w1, bias=bias3, act=ops.ACT_GELU_ERF, out=act), flops=fl)
        timeit('  fc1 fwd +c2 only', lambda: ops.linear_fwd(x, w1, c2=pre, out=act), flops=fl)
    timeit('fc2 dgrad x aux +colsum', lambda: ops.linear_dgrad(x, w2, act=ops.ACT_MUL_AUX, aux=pre, colsum=cs3, out=act), flops=fl)
    y = torch.empty((M, D), device='cuda', dtype=torch.bfloat16)
    timeit('fc2 fwd +bias+res', lambda: ops.linear_fwd(act, w2, bias=bias1, res=x, out=y), flops=fl)
    timeit('fc1 dgrad', lambda: ops.linear_dgrad(act, w1, out=y), flops=fl)
    wp = r(D, D)
    timeit('proj fwd +bias+res', lambda: ops.linear_fwd(x, wp, bias=bias1, res=x, out=y), flops=2.0 * M * D * D)
    timeit('proj dgrad', lambda: ops.linear_dgrad(x, wp, out=y), flops=2.0 * M * D * D)
    wq = r(3 * D, D)
    qkv = torch.empty((M, 3 * D), device='cuda', dtype=torch.bfloat16)
    timeit('qkv fwd +bias', lambda: ops.linear_fwd(x, wq, bias=torch.zeros(3 * D, device='cuda'), out=qkv), flops=2.0 * M * D * 3 * D)
    timeit('qkv dgrad', lambda: ops.linear_dgrad(qkv, wq, out=y), flops=2.0 * M * D * 3 * D)
    if 'wgrad' in want:
        dw = torch.zeros((3072, D), device='cuda')
        timeit('fc1 wgrad', lambda: ops.linear_wgrad(act, x, dw), flops=fl)
        dw2 = torch.zeros((D, 3072), device='cuda')
        timeit('fc2 wgrad', lambda: ops.linear_wgrad(x, act, dw2), flops=fl)
        dwq = torch.zeros((3 * D, D), device='cuda')
        timeit('qkv wgrad', lambda: ops.linear_wgrad(qkv, x, dwq), flops=2.0 * M * D * 3 * D)
        dwp = torch.zeros((D, D), device='cuda')
        timeit('proj wgrad', lambda: ops.linear_wgrad(y, x, dwp), flops=2.0 * M * D * D)
  ops.FORCE_TILE = 0
if 'layout' in want:
    # the same contraction with the weight stored [N][K] (k-major, ds_read_b128) vs [K][N] (transposing reads): is a transposed
    # bf16 shadow of the weights worth keeping for the data-gradient GEMMs?
    for name, K_, N_ in [('proj dgrad', 768, 768), ('qkv dgrad', 2304, 768), ('fc1 dgrad', 3072, 768), ('fc2 dgrad', 768, 3072)]:
        a_, wk, wn = r(M, K_), r(N_, K_), r(K_, N_)
        o_ = torch.empty((M, N_), device='cuda', dtype=torch.bfloat16)
        fl_ = 2.0 * M * K_ * N_
        timeit(f'{name} NT (W^T shadow)', lambda: ops.gemm(a_, wk, M, N_, K_, a_kmajor=True, b_kmajor=True, out=o_), flops=fl_)
        timeit(f'{name} NN (today)', lambda: ops.gemm(a_, wn, M, N_, K_, a_kmajor=True, b_kmajor=False, out=o_), flops=fl_)
        del a_, o_
if 'preproc' in want:
    from avt_amd.common.gpu_transforms import GpuClipTransform
    tf = GpuClipTransform('248-280', -1, 224, train=True)
    Bp = 64
    u8 = torch.randint(0, 256, (Bp, T, 256, 456, 3), device='cuda', dtype=torch.uint8)
    prm = [tf.draw(256, 456) for _ in range(Bp)]
    timeit(f'video_preproc {Bp}x{T} frames 256x456 -> 224^2', lambda: tf(u8, params=prm), bytes_=Bp * T * (256 * 456 * 3 + 3 * 224 * 224 * 4))
if 'sgd' in want:
    n = 396_120_000 // 64 * 64
    p_, g_, m_ = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    sh = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
    timeit('sgd 396M', lambda: ops.sgd_step(p_, g_, m_, sh, 1e-4, 0.9, 1e-6), bytes_=n * 26)
if 'blas' in want:
    # library reference: torch.matmul (hipBLASLt / rocBLAS heuristics) on the step's plain GEMM shapes, next to this repo's kernels
    import torch.nn.functional as F
    x768, x2304, x3072 = r(M, D), r(M, 3 * D), r(M, 4 * D)
    wq, w1, wp = r(3 * D, D), r(4 * D, D), r(D, D)
    bq = torch.rand(3 * D, device='cuda').to(torch.bfloat16)
    for name, fn, fl in [
        ('lib qkv fwd  x[M,768] Wq^T + b', lambda: F.linear(x768, wq, bq), 2.0 * M * D * 3 * D),
        ('lib fc1 dgrad dh[M,3072] W1', lambda: torch.matmul(x3072, w1), 2.0 * M * D * 4 * D),
        ('lib qkv dgrad dqkv[M,2304] Wq', lambda: torch.matmul(x2304, wq), 2.0 * M * D * 3 * D),
        ('lib proj dgrad dy[M,768] Wp', lambda: torch.matmul(x768, wp), 2.0 * M * D * D),
        ('lib fc1 wgrad dh^T x (bf16 out)', lambda: torch.matmul(x3072.t(), x768), 2.0 * M * D * 4 * D),
        ('lib qkv wgrad dqkv^T x (bf16 out)', lambda: torch.matmul(x2304.t(), x768), 2.0 * M * D * 3 * D),
        ('lib proj wgrad dy^T x (bf16 out)', lambda: torch.matmul(x768.t(), x768), 2.0 * M * D * D),
    ]:
        timeit(name, fn, flops=fl, iters=10)
    bq32 = bq.float()
    wT1, wTq = w1.t().contiguous(), wq.t().contiguous()
    g1, gq, gp = (torch.zeros(s, device='cuda') for s in [(4 * D, D), (3 * D, D), (D, D)])
    for name, fn, fl in [
        ('own qkv fwd', lambda: ops.linear_fwd(x768, wq, bias=bq32), 2.0 * M * D * 3 * D),
        ('own fc1 dgrad (W^T shadow, NT)', lambda: ops.linear_fwd(x3072, wT1), 2.0 * M * D * 4 * D),
        ('own qkv dgrad (W^T shadow, NT)', lambda: ops.linear_fwd(x2304, wTq), 2.0 * M * D * 3 * D),
        ('own proj dgrad (NN)', lambda: ops.linear_dgrad(x768, wp), 2.0 * M * D * D),
        ('own fc1 wgrad (fp32 accumulate)', lambda: ops.linear_wgrad(x3072, x768, g1), 2.0 * M * D * 4 * D),
        ('own qkv wgrad (fp32 accumulate)', lambda: ops.linear_wgrad(x2304, x768, gq), 2.0 * M * D * 3 * D),
        ('own proj wgrad (fp32 accumulate)', lambda: ops.linear_wgrad(x768, x768, gp), 2.0 * M * D * D),
    ]:
        timeit(name, fn, flops=fl, iters=10)
if 'sdpa' in want:
    # library reference for the ViT attention core: torch scaled_dot_product_attention (ROCm flash / mem-efficient backends)
    import torch.nn.functional as F
    qkv = r(M, 3 * D)
    q, k, v = (qkv.view(N, S, 3, H, 64)[:, :, i].permute(0, 2, 1, 3).contiguous().requires_grad_() for i in range(3))
    fl = 4.0 * N * H * S * S * 64
    timeit('lib sdpa fwd', lambda: F.scaled_dot_product_attention(q, k, v), flops=fl, iters=10)
    o = F.scaled_dot_product_attention(q, k, v)
    do = torch.randn_like(o)
    timeit('lib sdpa bwd', lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True), flops=2.5 * fl, iters=10)
    o2, l2 = ops.vit_attn_fwd(qkv, N, S, H)
    timeit('own vit_attn_fwd', lambda: ops.vit_attn_fwd(qkv, N, S, H), flops=fl, iters=10)
    db = torch.zeros(3 * D, device='cuda')
    timeit('own vit_attn_bwd (+dbias)', lambda: ops.vit_attn_bwd(qkv, o2, o2, l2, N, S, H, dbias=db), flops=2.5 * fl, iters=10)
