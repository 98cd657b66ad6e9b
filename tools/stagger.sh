#!/bin/bash
export AVT_HIP_LIB=${AVT_HIP_LIB:-$(pwd)/avt_amd/libavt_hip_lab.so}   # lab build (make -C avt_amd/csrc lab): the product library has no ablation / stagger switches
for st in 0 20000 40000 80000; do echo "STAGGER=$st"; AVT_GEMM_STAGGER=$st python tools/bench_square.py 2>&1 | grep -E "tile=256"; done
