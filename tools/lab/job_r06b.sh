#!/bin/bash
# r06b: the 8-rank gloo DDP tests; kernel traces of the current library at 64 / 16 / 3 clips per GPU (where the small batches lose against 256)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ddp_gpu.py -m gpu -x -q -k "eight" > gpurun_out/r06b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06b_pytest.log
tail -5 gpurun_out/r06b_pytest.log
for B in 64 16 3; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06b_$B -o t --output-format csv -- python bench.py --batch $B --steps 4 --warmup 3 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_r06b_$B.log 2>&1
  python tools/trace_summary.py gpurun_out/prof_r06b_$B/t_kernel_trace.csv 4 70 > gpurun_out/r06b_kernel_trace_B$B.txt 2>&1
  rm -rf gpurun_out/prof_r06b_$B
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06b_bench_B$B.json 2>/dev/null; cut -c1-330 gpurun_out/r06b_bench_B$B.json
done
