"""Whole training step with the first weight gradient of a step stored (avt_gemm_assign_bf16, the library's default since round 6) against every weight
gradient added into the re-zeroed buffer.   usage: python tools/lab/assign_ab.py {assign|accum} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd import ops
ops.ASSIGN_FIRST_WGRAD = sys.argv[1] == 'assign'
import bench
bench.main(sys.argv[2:])
