#!/bin/bash
# r04k: attention forward with software-pipelined V fragment reads (product) vs the previous build (libavt_prev.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or bench_size" > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log
for v in hip prev hip prev; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids
done | tee $O/kbench_attn.txt
timeout 600 python tools/lab/race_screen.py 40 > $O/race.txt 2>&1; tail -2 $O/race.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for v in hip prev hip prev; do
  AVT_HIP_LIB=$L/libavt_$v.so timeout 600 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v $O/bench_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:10s} {d['value']:8.1f} clips/s {d['ms_per_step']:8.2f} ms")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done | tee $O/bench.txt
