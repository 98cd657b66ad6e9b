#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aa; mkdir -p $O
timeout 300 python tools/lab/cu_contention.py 2>&1 | grep -v amdgpu.ids | tee $O/contention.txt
