#!/bin/bash
# r06s: split-K factors of the weight gradients at small batches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/lab/wgrad_split_sweep.py 3 8 16 32 > gpurun_out/r06s_wgrad_split_sweep.txt 2>&1; cat gpurun_out/r06s_wgrad_split_sweep.txt
