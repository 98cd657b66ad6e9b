#!/bin/bash
# round 5, session x: batch sweep of config 2 on the final library (SURVEY 8d's list 3 / 16 / 32 / 64, plus 128)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for B in 3 16 32 64 128; do
  timeout 600 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/r05x2_bench_B$B.json 2>/dev/null
  python - $B <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05x2_bench_B{sys.argv[1]}.json').read().strip().splitlines()[-1])
dk = d['roofline'].get('dominant_kernel', {})
print('B', sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['frac'], dk.get('frac'), dk.get('share_of_step_time'), d['host']['enqueue_ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05x2_small_batch.txt
