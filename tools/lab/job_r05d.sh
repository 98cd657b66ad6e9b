#!/bin/bash
# round 5, session c: 16-byte output stores in the ViT attention kernels (v_permlane16_swap) against the 8-byte stores
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or reproducible or bench_size" > gpurun_out/r05d_pytest.log 2>&1; tail -5 gpurun_out/r05d_pytest.log
for lib in libavt_narrow.so libavt_hip.so libavt_w2.so libavt_narrow.so libavt_hip.so libavt_w2.so; do
  echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib KB_BATCH=256 timeout 600 python tools/kbench.py attn 2>&1 | grep -v "^---\|amdgpu.ids"
done > gpurun_out/r05d_kbench.txt 2>&1
cat gpurun_out/r05d_kbench.txt
for lib in libavt_narrow.so libavt_hip.so libavt_w2.so libavt_narrow.so libavt_hip.so libavt_w2.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05d_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05d_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05d_steps.txt
