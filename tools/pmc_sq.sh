#!/bin/bash
# rocprofv3 SQ / GRBM / TCC counter passes over the bench step (counters in their own runs: --kernel-trace + --pmc only),
# summarised per (kernel, grid).  usage: tools/pmc_sq.sh TAG [bench args...]
TAG=${1:-r03a}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcsq_$TAG
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_IFETCH"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmcsq_$TAG/p$i -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-trace --no-also "$@" > gpurun_out/pmcsq_$TAG/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_sq_summary.py gpurun_out/pmcsq_$TAG 2 > gpurun_out/${TAG}_pmc_sq.txt 2>&1
head -60 gpurun_out/${TAG}_pmc_sq.txt
rm -rf gpurun_out/pmcsq_$TAG/p*/
