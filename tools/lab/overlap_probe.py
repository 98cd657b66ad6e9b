"""Can an HBM-bound kernel hide under a weight-gradient GEMM that leaves it some CUs?  gemm_w4_kernel with a split-K factor that fills 216 of the 256 CUs on
one stream, the folded LayerNorm backward on another, against the two back to back (full-width GEMM).  Sizes of the step at 256 clips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import ops, lib as _lib
M, D = 2560 * 197, 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x, dy, dres, dln = r(M, D), None, r(M, D), r(M, D)
rstd = torch.rand(M, device='cuda') + 0.5
sf = torch.stack([rstd, -0.1 * rstd], 1).contiguous()
cs = torch.zeros(D, device='cuda')
side = torch.cuda.Stream(priority=-1)          # the GEMM has to be placed first: the LayerNorm kernel would otherwise spread over every CU and keep the one-workgroup-per-CU GEMM out
def wgrad(dyt, xt, dw, splitk, stream):
    n, k = dw.shape
    need = _lib.load().avt_gemm_accum_workspace_bytes(n, k, M)
    ws = ops._wgrad_workspace(xt.device, need)
    _lib.call('avt_gemm_accum_bf16', dyt.data_ptr(), dyt.stride(0), xt.data_ptr(), xt.stride(0), dw.data_ptr(), dw.stride(0), n, k, M, splitk, 0, ws.data_ptr(), ws.numel(), stream.cuda_stream)
def timeit(name, fn, iters=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:70s} {e0.elapsed_time(e1) * 1e3 / iters:9.1f} us', flush=True)
main = torch.cuda.current_stream()
for name, N, sk_full, sk_part in [('fc1 wgrad (3072 x 768)', 3072, 0, 6), ('qkv wgrad (2304 x 768)', 2304, 0, 8)]:
    dyt = r(M, N)
    dw = torch.zeros((N, D), device='cuda')
    ws_side = None
    timeit(f'{name}: full width alone', lambda: wgrad(dyt, x, dw, sk_full, main))
    timeit(f'{name}: split {sk_part} (216 CUs) alone', lambda: wgrad(dyt, x, dw, sk_part, main))
    timeit('ln_bwd_folded alone', lambda: ops.layernorm_bwd_folded(dln, x, sf, dres=dres, colsum=cs))
    def seq():
        wgrad(dyt, x, dw, sk_full, main); ops.layernorm_bwd_folded(dln, x, sf, dres=dres, colsum=cs)
    timeit(f'{name}: full width, then ln_bwd_folded (one stream)', seq)
    def par(sk):
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        ev0 = torch.cuda.Event(); ev0.record(side)              # the side stream is at the GEMM's doorstep
        with torch.cuda.stream(side):
            wgrad(dyt, x, dw, sk, side)
        main.wait_event(ev0)                                    # ... only then may the LayerNorm kernel become eligible (the high-priority queue is served first)
        ops.layernorm_bwd_folded(dln, x, sf, dres=dres, colsum=cs)
        ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
    timeit(f'{name}: split {sk_part} on a side stream || ln_bwd_folded', lambda: par(sk_part))
    timeit(f'{name}: full width on a side stream || ln_bwd_folded', lambda: par(sk_full))
    for sk in (5, 4) if N == 3072 else (7, 6):
        timeit(f'{name}: split {sk} on a side stream || ln_bwd_folded', lambda: par(sk))
    del dyt
