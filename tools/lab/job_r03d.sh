cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 300 python tools/gemm_timeline.py 808 2562; AVT_GEMM_STAGGER=30000 timeout 300 python tools/gemm_timeline.py 2562) > gpurun_out/r03d_timeline.txt 2>&1
cat gpurun_out/r03d_timeline.txt
