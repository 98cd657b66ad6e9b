#!/bin/bash
# r06y: op-level test of the indirect seeds / device learning rate
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "indirect_seeds or sgd_step or skinny" > gpurun_out/r06y_pytest.log 2>&1; tail -15 gpurun_out/r06y_pytest.log
