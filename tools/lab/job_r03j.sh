cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 3 16; do
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_B$b -o t --output-format csv -- python bench.py --batch $b --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace > gpurun_out/prof_B$b.log 2>&1
python tools/trace_summary.py $(find gpurun_out/prof_B$b -name "*kernel_trace.csv" | head -1) 5 45 > gpurun_out/r03j_trace_B$b.txt 2>&1
rm -rf gpurun_out/prof_B$b
done
head -60 gpurun_out/r03j_trace_B3.txt
