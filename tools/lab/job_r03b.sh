cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so
(for st in 0 15000 30000 45000; do AVT_GEMM_STAGGER=$st timeout 300 python tools/lab/two_wg.py $( [ $st = 0 ] && echo 0 ) 2563 2562 1283; done) > gpurun_out/r03b_two_wg.txt 2>&1
cat gpurun_out/r03b_two_wg.txt
unset AVT_HIP_LIB
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest.log 2>&1; tail -5 gpurun_out/r03b_pytest.log
