#!/bin/bash
# r06q: full GPU suite on the round's library (first weight gradient of a step stored: avt_gemm_assign_bf16), whole-step A/B of that change, traces of the 3- / 64-clip steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06q_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06q_pytest.log; tail -12 gpurun_out/r06q_pytest.log
OUT=gpurun_out/r06q_assign.txt; : > $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in "3 40 8" "16 30 6" "64 15 4" "256 10 3"; do
  set -- $cfg
  for rep in 1 2; do
    for mode in accum assign; do
      timeout 600 python tools/lab/assign_ab.py $mode --batch $1 --steps $2 --warmup $3 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06q_ab.json 2>gpurun_out/r06q_ab.err; line "B=$1 $mode" gpurun_out/r06q_ab.json >> $OUT
    done
  done
done
cat $OUT
for B in 3 64; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06q_B$B -o t --output-format csv -- python bench.py --batch $B --steps 10 --warmup 0 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_r06q_B$B.log 2>&1
  python tools/trace_summary.py $(ls gpurun_out/prof_r06q_B$B/*/t_kernel_trace.csv gpurun_out/prof_r06q_B$B/t_kernel_trace.csv 2>/dev/null | head -1) 10 70 > gpurun_out/r06q_kernel_trace_B$B.txt 2>&1; head -12 gpurun_out/r06q_kernel_trace_B$B.txt
  rm -rf gpurun_out/prof_r06q_B$B
done
