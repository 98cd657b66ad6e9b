#!/bin/bash
# r04r: product library with the persistent GEMM: GPU test suite, bench (persist on), A/B against the lab library with AVT_GEMM_PERSIST=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for v in 15 0 15 0; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so AVT_GEMM_PERSIST=$v timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('persist=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
