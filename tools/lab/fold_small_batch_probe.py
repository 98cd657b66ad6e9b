"""Whole step at small batches with and without the LayerNorm fold (HipViT.fold_layernorm): below the persistent kernel's range the folded GELU epilogue of the
one-tile kernel looks its table up in global memory, and the fold's own small kernels (statistics, weight folding, folded weight gradients) are batch-independent."""
import os, sys, io, json, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from avt_amd.models.vit import HipViT
for B in [int(b) for b in os.environ.get("PROBE_BATCHES", "3,5,8,16").split(",")]:
    for fold in (True, False, True, False):
        HipViT.fold_layernorm = fold
        HipViT.fold_min_rows = 0              # (the product folds from 60000 token rows on; this probe decides by `fold` alone)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main(['--batch', str(B), '--steps', '30', '--warmup', '5', '--no-cpu-baseline', '--no-also', '--no-gemm-trace'])
        d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
        print(f'B {B:3d} fold {int(fold)}  {d["value"]:8.2f} clips/s  {d["ms_per_step"]:8.3f} ms  loss {d["config"].get("final_loss")}', flush=True)
