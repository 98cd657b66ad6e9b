#!/bin/bash
# r04e: single-pass attention backward after the NaN fix: parity; timing with and without the chunk barriers (nobar = wrong results, timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or bench_size or reductions_are_bit" > $O/pytest_attn.log 2>&1; tail -12 $O/pytest_attn.log
for v in hip bwd2ph nobar hip bwd2ph nobar; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids
done | tee $O/kbench_attn.txt
