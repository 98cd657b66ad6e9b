#!/bin/bash
# r06k: early optimizer step: tests, whole-step A/B at 3 / 16 / 64 / 256 clips per GPU; skinny kernel with XCD-paired tiles (sweep)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ddp_gpu.py -m gpu -q -x -k "early or two_rank" > gpurun_out/r06k_pytest.log 2>&1; tail -5 gpurun_out/r06k_pytest.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "torch_optimizer or grad_clip or train_net or checkpoint or trajectory or 20" > gpurun_out/r06k_pytest2.log 2>&1; tail -5 gpurun_out/r06k_pytest2.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "skinny" > gpurun_out/r06k_pytest3.log 2>&1; tail -3 gpurun_out/r06k_pytest3.log
OUT=gpurun_out/r06k_early_step.txt; : > $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}  enqueue {d['host']['enqueue_ms_per_step']} ms")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in "3 40 8" "16 30 6" "64 15 4" "256 10 3"; do
  set -- $cfg
  for rep in 1 2; do
    for mode in late early; do
      timeout 600 python tools/lab/early_step_ab.py $mode --batch $1 --steps $2 --warmup $3 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06k_ab.json 2>gpurun_out/r06k_ab.err; line "B=$1 $mode" gpurun_out/r06k_ab.json >> $OUT
    done
  done
done
cat $OUT
timeout 600 python tools/lab/skinny_sweep.py 30 > gpurun_out/r06k_skinny_sweep.txt 2>&1; cat gpurun_out/r06k_skinny_sweep.txt
