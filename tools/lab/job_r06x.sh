#!/bin/bash
# r06x: the round's final library on a fresh box: smoke, full GPU suite, default bench line (+ also), T = 15 / ViT-L lines, 256-clip kernel trace + PMC passes, 3-clip bench (eager / graph) + trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06x_smoke.log 2>&1; tail -2 gpurun_out/r06x_smoke.log
bash tools/gpu_session.sh r06x tests pmc
for mode in "" "--graph"; do
  timeout 600 python bench.py $mode --batch 3 --steps 60 --warmup 10 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06x_bench_B3$mode.json 2> /dev/null; cut -c1-330 gpurun_out/r06x_bench_B3$mode.json
done
for B in 16 64; do
  timeout 600 python bench.py --batch $B --steps 30 --warmup 6 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06x_bench_B$B.json 2> /dev/null; cut -c1-330 gpurun_out/r06x_bench_B$B.json
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06x_B3 -o t --output-format csv -- python bench.py --batch 3 --steps 10 --warmup 0 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_r06x_B3.log 2>&1
python tools/trace_summary.py $(ls gpurun_out/prof_r06x_B3/*/t_kernel_trace.csv gpurun_out/prof_r06x_B3/t_kernel_trace.csv 2>/dev/null | head -1) 10 70 > gpurun_out/r06x_kernel_trace_B3.txt 2>&1; head -8 gpurun_out/r06x_kernel_trace_B3.txt
rm -rf gpurun_out/prof_r06x_B3
