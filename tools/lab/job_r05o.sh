#!/bin/bash
# round 5o: attention backward, item tail before the last chunk's dQ products (libavt_hip.so) vs behind them (libavt_tf0.so); base = commit f12e8f3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o_attn.txt
: > $O
for lib in base tf0 hip; do
  echo "=== libavt_$lib.so" >> $O
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 1 2>&1 | grep "us per" >> $O
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 0 2>&1 | grep "us per" >> $O
done
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attn or attention or fold or bench_size or vit or reproducible" > gpurun_out/r05o_pytest.log 2>&1; tail -3 gpurun_out/r05o_pytest.log >> $O
for i in 1 2; do
  for lib in base tf0 hip; do
    AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-also --no-gemm-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
