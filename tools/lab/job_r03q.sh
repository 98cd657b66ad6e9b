cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03q -o r03q --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace > gpurun_out/prof_r03q.log 2>&1
python tools/trace_summary.py gpurun_out/prof_r03q/r03q_kernel_trace.csv 5 60 > gpurun_out/r03q_trace_summary.txt 2>&1; head -30 gpurun_out/r03q_trace_summary.txt
rm -f gpurun_out/prof_r03q/r03q_kernel_trace.csv
