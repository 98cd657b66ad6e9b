cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/lab/two_wg.py 0 2564 > gpurun_out/r03f_4w.txt 2>&1
cat gpurun_out/r03f_4w.txt
