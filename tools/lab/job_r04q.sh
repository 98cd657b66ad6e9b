#!/bin/bash
# r04q: persistent 8-phase GEMM vs gemm_8p_kernel: bit equality + time per launch (lab library, AVT_GEMM_PERSIST mask)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
export AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_${LIB:-hip_lab}.so
KINDS=${1:-1}
PC_RAGGED=1 timeout 300 python tools/lab/persist_check.py 700 $KINDS 2>&1 | grep -v amdgpu.ids | tee $O/check_ragged_$KINDS.txt
timeout 300 python tools/lab/persist_check.py 2560 $KINDS 2>&1 | grep -v amdgpu.ids | tee $O/check_$KINDS.txt
