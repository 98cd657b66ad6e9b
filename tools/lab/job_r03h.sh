cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_fc1_4w
bash tools/pmc_gemm.sh fc1_4w 504320 3072 768 NT 2564 > gpurun_out/r03g_pmc_fc1_4w.txt 2>&1
cat gpurun_out/r03g_pmc_fc1_4w.txt
rm -rf gpurun_out/pmc_fc1_4w
