#!/bin/bash
# r04n: early request of the epilogue's second operand (product) vs the previous build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm or linear or bench_size or epilogue or reproducible" > $O/pytest_gemm.log 2>&1; tail -4 $O/pytest_gemm.log
for v in hip prev hip prev; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/lab/two_wg.py 0 2>&1 | tail -1
done | tee $O/two_wg.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "g3 or config2 or g1 or g2" > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for v in hip prev hip prev; do
  AVT_HIP_LIB=$L/libavt_$v.so timeout 600 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v $O/bench_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:10s} {d['value']:8.1f} clips/s {d['ms_per_step']:8.2f} ms gemm {d['roofline']['dominant_kernel']['achieved']:.0f} loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done | tee $O/bench.txt
