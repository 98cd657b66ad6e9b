#!/bin/bash
# round 5, session zf: kernel trace of the 3-clip step on the final routing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=3
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_zf -o zf --output-format csv -- python bench.py --batch $B --steps 5 --warmup 3 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_zf.log 2>&1
python tools/trace_summary.py gpurun_out/prof_zf/zf_kernel_trace.csv 8 70 > gpurun_out/r05zf_kernel_trace_B3.txt 2>&1
rm -f gpurun_out/prof_zf/zf_kernel_trace.csv
head -60 gpurun_out/r05zf_kernel_trace_B3.txt
