#!/bin/bash
# round 5, session i: small-batch sweep on the round-5 library (SURVEY 8d's config-2 batch list) 
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for B in 3 16 32 64 128; do
  timeout 600 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-also --trace-steps 3 > gpurun_out/r05i_bench_B$B.json 2>/dev/null
  python - $B <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05i_bench_B{sys.argv[1]}.json').read().strip().splitlines()[-1]); r = d['roofline']
print('B', sys.argv[1], d['value'], d['ms_per_step'], r['frac'], r.get('dominant_kernel', {}).get('frac'), r.get('dominant_kernel', {}).get('share_of_step_time'), d['host']['enqueue_ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05i_small_batch.txt
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05i -o r05i --output-format csv -- python bench.py --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_r05i.log 2>&1
python tools/trace_summary.py gpurun_out/prof_r05i/r05i_kernel_trace.csv 5 70 > gpurun_out/r05i_trace_B64.txt 2>&1; head -60 gpurun_out/r05i_trace_B64.txt
rm -f gpurun_out/prof_r05i/r05i_kernel_trace.csv
