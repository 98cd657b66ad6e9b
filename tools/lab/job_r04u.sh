#!/bin/bash
# r04u: persistent GEMM, first (static) version vs current, each against gemm_8p_kernel in its own process
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
for lib in p1 hip_lab p1 hip_lab; do
  echo "== lib $lib" | tee -a $O/cmp.txt
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/persist_check.py 2560 15 2>&1 | grep -v amdgpu.ids | tee -a $O/cmp.txt
done
