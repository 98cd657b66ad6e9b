#!/bin/bash
# r04p: patch embedding as an implicit GEMM (avt_patch_embed_fwd / _wgrad) vs im2col + GEMM: parity, per-launch time, whole step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "patch_embed or im2col" > $O/pytest_ops.log 2>&1; tail -25 $O/pytest_ops.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/kbench_patch.txt
import torch, sys
sys.path.insert(0, '.')
from avt_amd import ops
N, D, S = 2560, 768, 197
frames = torch.rand((N, 3, 224, 224), device='cuda') * 2 - 1
wb = (torch.randn((D, 768), device='cuda') * 0.02).to(torch.bfloat16)
R = (torch.randn((S, D), device='cuda') * 0.1).to(torch.bfloat16)
dx = (torch.randn((N * S, D), device='cuda')).to(torch.bfloat16)
dw = torch.zeros((D, 768), device='cuda')
def t(name, fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); print(f'{name:40s} {e0.elapsed_time(e1) * 1e3 / it:9.1f} us', flush=True)
patches = ops.im2col_patch16(frames)
t('im2col16', lambda: ops.im2col_patch16(frames))
t('gemm on patches (+posres)', lambda: ops.gemm(patches, wb, N * S, D, 768, res=R, res_period=S))
t('implicit fwd (+posres)', lambda: ops.patch_embed_fwd(frames, wb, R, D))
t('wgrad on patches', lambda: ops.linear_wgrad(dx, patches, dw))
t('implicit wgrad', lambda: ops.patch_embed_wgrad(dx, frames, dw))
PY
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "g3 or config2 or vitl or g7 or smoke" > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for v in 1 0 1 0; do
  timeout 600 python - $v <<'PY' 2>/dev/null | tail -1
import sys, json, io, contextlib
sys.path.insert(0, '.')
from avt_amd import ops
ops.IMPLICIT_PATCH_EMBED = bool(int(sys.argv[1]))
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(['--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-also'])
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"implicit={sys.argv[1]} {d['value']:8.1f} clips/s {d['ms_per_step']:8.2f} ms loss {d['config']['final_loss']}")
PY
done | tee $O/bench.txt
