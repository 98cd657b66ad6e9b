#!/bin/bash
# round 5, session s: fragment-major GELU' (AVT_AUX_FRAG) A/B -- fc1 forward / fc2 data gradient per launch, then the whole step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in libavt_hip.so libavt_auxfrag.so libavt_hip.so libavt_auxfrag.so; do echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib python tools/lab/lnf_bench.py 2>&1 | grep -v amdgpu.ids | grep "fc1 fwd\|fc2 dgrad\|qkv fwd plain"; done | tee gpurun_out/r05s_auxfrag_gemms.txt
for lib in libavt_hip.so libavt_auxfrag.so libavt_hip.so libavt_auxfrag.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05s_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05s_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
done | tee gpurun_out/r05s_steps.txt
