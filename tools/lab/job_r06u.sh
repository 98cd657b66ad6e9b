#!/bin/bash
# r06u: 3 clips per GPU: the 288-tile GEMMs (fc1 forward, fc2 data gradient) as one full round on the 8-phase kernel + the remaining rows on the 64 x 64 ring
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06u_row_split.txt; : > $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for rep in 1 2 3; do
  for mode in one split; do
    timeout 600 python tools/lab/row_split_ab.py $mode --batch 3 --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06u_ab.json 2>gpurun_out/r06u_ab.err; line "B=3 $mode" gpurun_out/r06u_ab.json >> $OUT
  done
done
cat $OUT; tail -3 gpurun_out/r06u_ab.err
