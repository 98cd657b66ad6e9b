#!/bin/bash
# r06c: roll-out with gradients vs golden G12; the RCCL C-ABI entry points at world size 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -k "g12 or c_abi or rccl_executes or g6" > gpurun_out/r06c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06c_pytest.log
tail -25 gpurun_out/r06c_pytest.log
