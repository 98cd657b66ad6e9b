#!/bin/bash
# round 5n: attention kernels with untracked strip loads (no compiler-inserted vmcnt(0)); A/B against the library of commit f12e8f3 (libavt_base.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n_attn.txt
: > $O
for lib in base hip; do
  echo "=== libavt_$lib.so" >> $O
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 1 >> $O 2>&1
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 300 python tools/lab/attn_timeline.py 2560 0 2>&1 | grep "us per" >> $O
done
echo "=== lab library (stamps)" >> $O
AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so timeout 300 python tools/lab/attn_timeline.py 2560 1 >> $O 2>&1
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attn or attention or fold or bench_size or vit" > gpurun_out/r05n_pytest.log 2>&1; tail -3 gpurun_out/r05n_pytest.log >> $O
for i in 1 2; do
  for lib in base hip; do
    AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-also --no-gemm-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])" >> $O
  done
done
cat $O
