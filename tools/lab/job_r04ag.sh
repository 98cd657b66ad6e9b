#!/bin/bash
# r04ag: two ranks sharing the one GPU (gloo: functional check of the N > 1 path with the persistent GEMM active), 64 clips each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ag; mkdir -p $O
timeout 900 python bench.py --gpus 2 --backend gloo --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-also > $O/two_ranks.json 2> $O/two_ranks.err; echo rc=$?; tail -c 1200 $O/two_ranks.json; tail -3 $O/two_ranks.err
timeout 600 python bench.py --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | cut -c1-300
