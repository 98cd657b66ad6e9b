#!/bin/bash
# round 5, session m: placement of the fold's row-statistics loads in the persistent kernel's last K iteration (A/B), fold variants next to the plain epilogues
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "fold or statistics or scaled or gelu" > gpurun_out/r05m_pytest.log 2>&1; tail -3 gpurun_out/r05m_pytest.log
for lib in libavt_oldfold.so libavt_hip.so libavt_oldfold.so libavt_hip.so; do echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib python tools/lab/lnf_bench.py 2>&1 | grep -v amdgpu.ids | grep "fwd\|dgrad"; done | tee gpurun_out/r05m_fold_gemms.txt
for lib in libavt_oldfold.so libavt_hip.so libavt_oldfold.so libavt_hip.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05m_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05m_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05m_steps.txt
