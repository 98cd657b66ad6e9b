#!/bin/bash
# round 5, session z: tile choice of the ViT GEMMs at small batches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lab/small_batch_gemm_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05z_small_batch_gemm_sweep.txt
