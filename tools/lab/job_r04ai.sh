#!/bin/bash
# r04ai: GELU table offsets in 8 instead of 9 vector instructions per pair: tests + whole step old vs new library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ai; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gelu or persistent or epilogue" 2>&1 | tail -3
B="--steps 10 --warmup 3 --no-cpu-baseline --no-also"
for rep in 1 2 3; do for v in old hip; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$v.so timeout 600 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'])" | tee -a $O/ab.txt
done; done
