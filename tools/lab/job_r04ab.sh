#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ab; mkdir -p $O
timeout 300 python tools/lab/k4_check.py 700 2>&1 | grep -v amdgpu.ids | tee $O/k4_ragged.txt
timeout 300 python tools/lab/k4_check.py 2560 2>&1 | grep -v amdgpu.ids | tee $O/k4.txt
