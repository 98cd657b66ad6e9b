#!/bin/bash
# round 5, session f: LayerNorm fold -- op-level and model-level parity, then the full GPU suite, then bench A/B fold on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fold or statistics or scaled" > gpurun_out/r05f_pytest_ops.log 2>&1; tail -25 gpurun_out/r05f_pytest_ops.log
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r05f_pytest_all.log 2>&1; tail -15 gpurun_out/r05f_pytest_all.log
for fold in 1 0 1 0; do
  AVT_FOLD_LN=$fold timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05f_bench_tmp.json 2>gpurun_out/r05f_bench_tmp.err || tail -5 gpurun_out/r05f_bench_tmp.err
  python - $fold <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05f_bench_tmp.json').read().strip().splitlines()[-1]); print('fold', sys.argv[1], d['value'], d['ms_per_step'], d['config']['final_loss'], flush=True)
PY
done | tee gpurun_out/r05f_steps.txt
