#!/bin/bash
# round 5q: start stagger of the attention kernels' persistent workgroups (libavt_stgN.so: workgroup b sleeps (b mod 16) x N x 64 cycles first)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q_attn_stagger.txt
: > $O
for lib in hip stg8 stg17 stg34 hip; do
  [ -f avt_amd/libavt_$lib.so ] || continue
  echo "=== libavt_$lib.so" >> $O
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 1 2>&1 | grep "us per" >> $O
done
cat $O
