#!/bin/bash
# r04s: scalar-atomic probe + kernel trace of the step with the persistent GEMM
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 60 tools/probe/satomic_probe 2>&1 | tee $O/satomic.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace --no-also > $O/prof.log 2>&1
python tools/trace_summary.py $O/prof/t_kernel_trace.csv 5 60 > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt
rm -f $O/prof/t_kernel_trace.csv
