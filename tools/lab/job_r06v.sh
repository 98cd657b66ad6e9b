#!/bin/bash
# r06v: the final library's default line on another box of the pool (box-to-box spread), plus the 3-clip lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06v_bench.json 2> gpurun_out/r06v_bench.err; cut -c1-400 gpurun_out/r06v_bench.json
for mode in "" "--graph"; do
  timeout 600 python bench.py $mode --batch 3 --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06v_bench_B3$mode.json 2> /dev/null; cut -c1-330 gpurun_out/r06v_bench_B3$mode.json
done
