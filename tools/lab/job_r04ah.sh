#!/bin/bash
# r04ah: whole step, lab library, persistent GEMM for K <= 1024 / 2304 / every K (final kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ah; mkdir -p $O
export AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for rep in 1 2; do for km in 0 1024 2304 99999; do
  P=15; [ $km = 0 ] && P=0
  AVT_GEMM_PERSIST=$P AVT_GEMM_PERSIST_KMAX=$km timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('persist=$P kmax=$km', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
done; done
