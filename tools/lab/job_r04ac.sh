#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ac; mkdir -p $O
for lib in ${LIBS:-hip ln3 hip ln3}; do echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 200 python tools/lab/ln_check.py 2>&1 | grep -v amdgpu.ids; done | tee $O/ln.txt
AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip.so timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "layernorm or column_sum" 2>&1 | tail -3
