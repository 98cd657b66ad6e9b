"""Weight gradients at small batches: us per call (kernel + ordered slab reduce) of the 4-wave kernel for every split-K factor, against the automatic one
(pick_splitk fills the 256 CUs; at 3 clips a split is then 26 K-steps long and the slabs are 14x the gradient).   usage: python tools/lab/wgrad_split_sweep.py [clips ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avt_amd import lib
D = 768
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
ws = torch.empty(1 << 30, device='cuda', dtype=torch.uint8)
for clips in [int(a) for a in sys.argv[1:]] or [3, 8, 16]:
    M = clips * 1970
    print(f'== {clips} clips: reduction over M = {M} rows; us per call incl. the slab reduce (200 calls, HIP events); splitk 0 = automatic')
    for name, (n_dy, n_x) in (('proj', (D, D)), ('qkv', (3 * D, D)), ('fc1', (4 * D, D)), ('fc2', (D, 4 * D)), ('head c_fc (30 x clips/3 rows)', (2048, 8192))):
        K = M if not name.startswith('head') else 10 * clips
        dy, x = r(K, n_dy), r(K, n_x)
        dw = torch.zeros((n_dy, n_x), device='cuda')
        row = f'{name:32s} dW {n_dy:5d} x {n_x:5d} '
        for sk in (0, 1, 2, 3, 4, 5, 6, 8, 12):
            def call(entry='avt_gemm_accum_bf16'):
                lib.call(entry, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), n_dy, n_x, K, sk, 0,
                         ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            try:
                for _ in range(5):
                    call()
            except lib.AvtHipError:
                row += f'  {sk}: -'
                continue
            torch.cuda.synchronize()
            res = []
            for entry in ('avt_gemm_accum_bf16', 'avt_gemm_assign_bf16'):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    call(entry)
                e1.record(); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / 200)
            row += f'  {sk}: {res[0]:5.1f}/{res[1]:5.1f}'
        print(row + '   (accumulate / assign)', flush=True)
