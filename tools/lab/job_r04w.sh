#!/bin/bash
# r04w: whole step, lab library: persistent GEMM off / on (K <= 1024)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
export AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for rep in 1 2 3; do
for cfg in ${CFGS:-"0 1024" "15 1024"}; do
  set -- $cfg
  AVT_GEMM_PERSIST=$1 AVT_GEMM_PERSIST_KMAX=$2 timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('persist=$1 kmax=$2', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
done; done
