#!/bin/bash
# r04af: clips per GPU chosen so that the GEMM tile grids fill whole rounds of the 256 CUs (266: 2047 row tiles) against 256 (1970 row tiles)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04af; mkdir -p $O
for rep in 1 2; do for b in 256 266 277; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --batch $b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch=$b', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/batch.txt
done; done
