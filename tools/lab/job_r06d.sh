#!/bin/bash
# r06d: patch rows from the input pipeline, RCCL C ABI at world 1, roll-out with gradients
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q -k "patch_rows or preproc or jitter or train_net or g6b or g12 or c_abi or rccl_executes" > gpurun_out/r06d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06d_pytest.log
tail -30 gpurun_out/r06d_pytest.log
