"""Lab: a GEMM launch while `nwg` CUs are held by another kernel on a second stream (tools/lab/cu_hog.hip) -- the backward GEMMs of a
data-parallel step run while RCCL's kernels occupy CUs.  One tile per workgroup (gemm_8p_kernel, tile 808) against the persistent kernel with
dynamic tile tickets (tile 809): a persistent grid with a STATIC tile assignment would wait for the held CUs and take twice as long.
usage: python tools/lab/cu_contention.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from avt_amd import ops
hog = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libcu_hog.so'))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device='cuda', dtype=torch.int32)
side = torch.cuda.Stream()
M = 2560 * 197
r = lambda *s: (torch.rand(s, device='cuda') * 2 - 1).to(torch.bfloat16)
x = r(M, 768)
for name, N, kw in [('qkv fwd K=768 N=2304 (bias)', 2304, dict(bias=torch.rand(2304, device='cuda'))),
                    ('proj fwd K=768 N=768 (bias + res)', 768, dict(bias=torch.rand(768, device='cuda'), res=r(M, 768)))]:
    W = r(N, 768)
    out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
    ref = None
    for nwg in (0, 16, 32, 64):
        row = []
        for tile in (808, 809):
            best = 1e30
            for _ in range(5):
                torch.cuda.synchronize()
                ops.linear_fwd(x, W, out=out, tile=tile, **kw)                        # (clocks up: the timed launch follows a busy chip, as in a step)
                if nwg:
                    hog.cu_hog(nwg, 8000, sink.data_ptr(), side.cuda_stream)      # holds its CUs for 8 ms
                torch.cuda._sleep(200000)                                               # let the hog workgroups settle first
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.linear_fwd(x, W, out=out, tile=tile, **kw); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref)
            row.append(best)
        print(f'{name:36s} {nwg:3d} CUs held: one tile per workgroup {row[0]:8.1f} us   persistent {row[1]:8.1f} us   (ideal x{256 / (256 - nwg):.3f})', flush=True)
