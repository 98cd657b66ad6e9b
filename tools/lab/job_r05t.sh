#!/bin/bash
# round 5, session t: timing-only ablations of the persistent GEMM's epilogues (AVT_PK_ABL: 1 = no table gathers, 2 = stores stay in L2, 4 = no LDS patch round trip, 7 = all)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in libavt_hip.so libavt_abl1.so libavt_abl2.so libavt_abl4.so libavt_abl7.so libavt_hip.so; do echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib python tools/lab/epi_abl_bench.py 2>&1 | grep " us"; done | tee gpurun_out/r05t_epilogue_ablations.txt
