#!/bin/bash
# PMC passes for one GEMM config (counters in their own runs, kernel-trace only). usage: pmc_gemm.sh TAG M N K layout tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
            "GRBM_GUI_ACTIVE GRBM_COUNT FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmc_$TAG/$n -o p --output-format csv -- python tools/one_gemm.py "$@" > gpurun_out/pmc_$TAG/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_$TAG/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        if 'gemm' not in k: continue
        tot[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[(k, row['Counter_Name'])] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} {v / cnt[(k, c)]:16.1f} (avg per dispatch over {cnt[(k, c)]})')
PY
