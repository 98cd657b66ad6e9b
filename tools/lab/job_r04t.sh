#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
for lib in ${LIBS:-hip_lab}; do
echo "== lib $lib" | tee -a $O/timeline.txt
AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/persist_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O/timeline.txt
done
