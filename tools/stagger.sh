#!/bin/bash
for st in 0 20000 40000 80000; do echo "STAGGER=$st"; AVT_GEMM_STAGGER=$st python tools/bench_square.py 2>&1 | grep -E "tile=256"; done
