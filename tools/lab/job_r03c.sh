cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 300 python tools/gemm_timeline.py 808 2562; AVT_GEMM_STAGGER=30000 timeout 300 python tools/gemm_timeline.py 2562) > gpurun_out/r03c_timeline.txt 2>&1
cat gpurun_out/r03c_timeline.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "g8b or vitl_full_depth_backward" > gpurun_out/r03c_pytest.log 2>&1; tail -12 gpurun_out/r03c_pytest.log
