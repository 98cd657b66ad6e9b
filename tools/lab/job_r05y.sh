cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for B in 3 16; do
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b$B -o b$B --output-format csv -- python bench.py --batch $B --steps 5 --warmup 3 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_b$B.log 2>&1
python tools/trace_summary.py gpurun_out/prof_b$B/b${B}_kernel_trace.csv 5 70 > gpurun_out/r05x2_kernel_trace_B$B.txt 2>&1
rm -f gpurun_out/prof_b$B/b${B}_kernel_trace.csv
done
head -50 gpurun_out/r05x2_kernel_trace_B3.txt
