#!/bin/bash
# r06w: the skinny kernel with two row tiles (33-64 rows: the head at 3 clips x 15 frames = 45 rows, BASELINE config 4 at the reference's batch): parity, sweep, whole-step A/B at T = 15
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "skinny or gemm_plain or epilogue" > gpurun_out/r06w_pytest.log 2>&1; tail -4 gpurun_out/r06w_pytest.log
OUT=gpurun_out/r06w_skinny64.txt; : > $OUT
timeout 600 python tools/lab/skinny_sweep.py 45 >> $OUT 2>&1
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:24s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in "3 15" "4 15" "6 10" "3 10"; do
  set -- $cfg
  for rep in 1 2; do
    for mode in tiles64 skinny; do
      timeout 600 python tools/lab/skinny_ab.py $mode --batch $1 --frames $2 --steps 40 --warmup 8 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06w_ab.json 2>gpurun_out/r06w_ab.err; line "B=$1 T=$2 $mode" gpurun_out/r06w_ab.json >> $OUT
    done
  done
done
cat $OUT
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "config4 or T15 or bench_size or g2b or route" > gpurun_out/r06w_pytest_model.log 2>&1; tail -4 gpurun_out/r06w_pytest_model.log
