#!/bin/bash
# r06j: the skinny GEMM kernel (<= 32 output rows): parity tests, per-shape sweep, whole-step A/B at 1 / 2 / 3 clips per GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "skinny or gemm_plain or epilogue or deterministic_wgrad" > gpurun_out/r06j_pytest.log 2>&1; tail -5 gpurun_out/r06j_pytest.log
OUT=gpurun_out/r06j_skinny_gemm.txt; : > $OUT
timeout 600 python tools/lab/skinny_sweep.py 30 >> $OUT 2>&1
timeout 600 python tools/lab/skinny_sweep.py 10 >> $OUT 2>&1
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:22s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}  enqueue {d['host']['enqueue_ms_per_step']} ms")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for B in 3 1 2; do
  for rep in 1 2; do
    for mode in tiles64 skinny; do
      timeout 600 python tools/lab/skinny_ab.py $mode --batch $B --steps 40 --warmup 8 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06j_ab.json 2>gpurun_out/r06j_ab.err; line "B=$B $mode" gpurun_out/r06j_ab.json >> $OUT
    done
  done
done
cat $OUT
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "bench_size or route or golden or g2 or g8" > gpurun_out/r06j_pytest_model.log 2>&1; tail -5 gpurun_out/r06j_pytest_model.log
