#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ae; mkdir -p $O
timeout 600 python tools/lab/head_gemm_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/head.txt
