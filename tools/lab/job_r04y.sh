#!/bin/bash
# r04y: whole step with several builds of gemm_persist.hip on one box (tools/lab/build_persist_variant.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for rep in 1 2; do
for v in ${VARIANTS:-vA vB vC}; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$v.so timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
done; done
AVT_GEMM_PERSIST=0 AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_vB.so timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('off', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
