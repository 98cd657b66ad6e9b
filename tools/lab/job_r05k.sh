#!/bin/bash
# round 5, session j: the persistent GEMM for every K (PK_KMAX 4096) against K <= 1024 on the round-5 library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in libavt_hip.so libavt_fullpf.so libavt_hip.so libavt_fullpf.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05k_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05k_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05k_steps.txt
