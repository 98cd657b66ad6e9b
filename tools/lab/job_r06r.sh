#!/bin/bash
# r06r: the all-token ViT blocks' weight gradients on a second stream (small-batch route): whole-step A/B at 3 / 8 / 12 / 16 / 20 / 24 clips per GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06r_wgrad_side.txt; : > $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in "16 30 6" "3 40 8" "8 40 8" "12 30 6" "20 30 6" "24 30 6"; do
  set -- $cfg
  for rep in 1 2; do
    for mode in one side; do
      timeout 600 python tools/lab/wgrad_side_ab.py $mode --batch $1 --steps $2 --warmup $3 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06r_ab.json 2>gpurun_out/r06r_ab.err; line "B=$1 $mode" gpurun_out/r06r_ab.json >> $OUT
    done
  done
done
cat $OUT; tail -3 gpurun_out/r06r_ab.err
