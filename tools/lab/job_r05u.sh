#!/bin/bash
# round 5, session u: the fragment-major GELU' in the product (ABI 7) -- new tests, the GEMM / model suites, then the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "fragment" > gpurun_out/r05u_pytest_new.log 2>&1; tail -15 gpurun_out/r05u_pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05u_pytest_all.log 2>&1; tail -4 gpurun_out/r05u_pytest_all.log
for i in 1 2; do
  timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05u_bench_tmp.json 2>/dev/null
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05u_bench_tmp.json').read().strip().splitlines()[-1]); print('product', d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
done | tee gpurun_out/r05u_steps.txt
