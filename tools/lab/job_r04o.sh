#!/bin/bash
# r04o: column-strip width of the 8-phase kernel re-swept now that the output stores are non-temporal (lab build, AVT_GEMM_STRIP)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
for st in auto 0 3 4 6 auto 0 4; do
  echo "== strip $st"
  if [ $st = auto ]; then AVT_HIP_LIB=$L/libavt_hip_lab.so KB_BATCH=256 timeout 300 python tools/lab/two_wg.py 0 2>&1 | tail -1
  else AVT_GEMM_STRIP=$st AVT_HIP_LIB=$L/libavt_hip_lab.so KB_BATCH=256 timeout 300 python tools/lab/two_wg.py 0 2>&1 | tail -1; fi
done | tee $O/strips.txt
timeout 300 bash tools/scale_sweep.sh 1 3 2>&1 | tail -3
