#!/bin/bash
# round 5, session b: de-waterfalled epilogue stores (T20) and patch-free swap stores (T21) in the persistent GEMM: parity, per-launch, whole step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "persistent or gemm or gelu" > gpurun_out/r05b_pytest_gemm.log 2>&1; tail -5 gpurun_out/r05b_pytest_gemm.log
for lib in libavt_base.so libavt_sw0.so libavt_sw1.so libavt_sw2.so libavt_sw4.so libavt_hip.so libavt_base.so libavt_hip.so; do
  echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib KB_BATCH=256 timeout 600 python tools/kbench.py gemm 2>&1 | grep -v "^---\|amdgpu.ids" 
done > gpurun_out/r05b_kbench.txt 2>&1
cat gpurun_out/r05b_kbench.txt
for lib in libavt_base.so libavt_sw0.so libavt_hip.so libavt_base.so libavt_sw0.so libavt_hip.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05b_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05b_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05b_steps.txt
