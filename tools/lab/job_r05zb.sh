#!/bin/bash
# round 5, session zb: same-box A/B of the small-shape tile rules: libavt_oldrule.so (-DAVT_OLD_SMALL_TILE_RULE) against the product, whole step at 256 / 3 / 8 clips
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in libavt_oldrule.so libavt_hip.so libavt_oldrule.so libavt_hip.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05zb_tmp.json 2>/dev/null
  python - $lib 256 <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05zb_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'B', sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
done | tee gpurun_out/r05zb_ab.txt
for B in 3 8; do for lib in libavt_oldrule.so libavt_hip.so libavt_oldrule.so libavt_hip.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 600 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r05zb_tmp.json 2>/dev/null
  python - $lib $B <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05zb_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'B', sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
done; done | tee -a gpurun_out/r05zb_ab.txt
