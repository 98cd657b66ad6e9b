#!/bin/bash
# r04h: single-pass attention backward with software-pipelined fragment reads (product) vs the same without (bwd1np) vs two-phase (bwd2ph):
# parity, race screen, per-launch time, in-step per-kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or bench_size or reductions_are_bit" > $O/pytest_attn.log 2>&1; tail -4 $O/pytest_attn.log
for v in hip bwd1np bwd2ph hip bwd1np bwd2ph; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids
done | tee $O/kbench_attn.txt
timeout 600 python tools/lab/race_screen.py 60 > $O/race.txt 2>&1; tail -3 $O/race.txt
for v in hip bwd2ph; do
  AVT_HIP_LIB=$L/libavt_$v.so timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o t --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace --no-also > $O/prof_$v.log 2>&1
  python tools/trace_summary.py $O/prof_$v/t_kernel_trace.csv 5 40 > $O/trace_$v.txt 2>&1; head -14 $O/trace_$v.txt
  rm -rf $O/prof_$v
done
