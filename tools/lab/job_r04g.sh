#!/bin/bash
# r04g: in-step per-kernel times (rocprofv3 kernel trace, 3 steps) with the single-pass attention backward (product) and the two-phase one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
for v in hip bwd2ph; do
  AVT_HIP_LIB=$L/libavt_$v.so timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o t --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace --no-also > $O/prof_$v.log 2>&1
  python tools/trace_summary.py $O/prof_$v/t_kernel_trace.csv 5 40 > $O/trace_$v.txt 2>&1; head -16 $O/trace_$v.txt
  rm -f $O/prof_$v/t_kernel_trace.csv
done
