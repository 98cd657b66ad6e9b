#!/bin/bash
# r06zb: after the registry change (Python only): the tests that exercise it on the GPU, and the 3-clip lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q -x -k "first_weight_gradient or deterministic_wgrad or captured or config2 or reproducible or two_rank or torch_optimizer or grad_clip or multi_crop or rollout or g12 or g6" > gpurun_out/r06zb_pytest.log 2>&1; tail -4 gpurun_out/r06zb_pytest.log
for mode in "" "--graph"; do
  timeout 600 python bench.py $mode --batch 3 --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06zb_bench_B3$mode.json 2> /dev/null; python - gpurun_out/r06zb_bench_B3$mode.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['config']['launch'][:8], d['value'], d['ms_per_step'], d['roofline']['frac'], d['host'])
PY
done
