cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_timeline.py 808 2564 > gpurun_out/r03g_timeline.txt 2>&1
cat gpurun_out/r03g_timeline.txt
mkdir -p gpurun_out/pmc_fc1_8p gpurun_out/pmc_fc1_4w
bash tools/pmc_gemm.sh fc1_8p 504320 3072 768 NT 808 > gpurun_out/r03g_pmc_fc1_8p.txt 2>&1
bash tools/pmc_gemm.sh fc1_4w 504320 3072 768 NT 2564 > gpurun_out/r03g_pmc_fc1_4w.txt 2>&1
cat gpurun_out/r03g_pmc_fc1_8p.txt gpurun_out/r03g_pmc_fc1_4w.txt
rm -rf gpurun_out/pmc_fc1_8p gpurun_out/pmc_fc1_4w
