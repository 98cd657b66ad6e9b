#!/bin/bash
# round 5r: start stagger of the persistent GEMM's workgroups (libavt_pksN.so: spread over N cycles), whole-step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r_pk_stagger.txt
: > $O
for i in 1 2; do
  for lib in hip pks30000 pks60000; do
    AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
