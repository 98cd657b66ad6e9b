#!/bin/bash
# round 5, session w: do weight gradients on a second stream fill the tails of the data-gradient chain?  (tools/lab/tail_overlap_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lab/tail_overlap_probe.py 2>&1 | grep " us" | tee gpurun_out/r05w_tail_overlap.txt
