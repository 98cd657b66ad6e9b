#!/bin/bash
# round 5p: timing-only ablations of the single-pass attention backward (AVT_ATTN_ABL bits; WRONG results by construction)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p_attn_ablations.txt
: > $O
for lib in hip abl32 abl64 abl96 hip; do
  [ -f avt_amd/libavt_$lib.so ] || continue
  echo "=== libavt_$lib.so" >> $O
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 1 2>&1 | grep "scaled 1" >> $O
done
cat $O
