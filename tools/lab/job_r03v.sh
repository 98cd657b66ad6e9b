cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_fc2_8p gpurun_out/pmc_fc2_w4n
bash tools/pmc_gemm.sh fc2_8p 504320 768 3072 NT 808 2>&1 | grep -E "gemm|MFMA_BUSY|GRBM_GUI|TCC|FETCH|WAIT_INST_ANY|WAVE_CYCLES|LDS_IDX|INSTS_VALU"
bash tools/pmc_gemm.sh fc2_w4n 504320 768 3072 NT 2566 2>&1 | grep -E "gemm|MFMA_BUSY|GRBM_GUI|TCC|FETCH|WAIT_INST_ANY|WAVE_CYCLES|LDS_IDX|INSTS_VALU"
rm -rf gpurun_out/pmc_fc2_8p gpurun_out/pmc_fc2_w4n
