#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ad; mkdir -p $O
timeout 600 python tools/lab/host_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/host.txt
