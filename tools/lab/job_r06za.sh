#!/bin/bash
# r06za: where the LayerNorm fold starts to pay, re-measured after round 6's changes to the unfolded route (assign-first weight gradients)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PROBE_BATCHES=16,24,32,40,48,64 timeout 2400 python tools/lab/fold_small_batch_probe.py 2>/dev/null | grep "^B " > gpurun_out/r06za_fold_threshold.txt; cat gpurun_out/r06za_fold_threshold.txt
