#!/bin/bash
# latency-oriented PMC passes. usage: pmc2.sh TAG M N K layout tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/pmc2_$TAG
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
            "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmc2_$TAG/p$i -o p --output-format csv -- python tools/one_gemm.py "$@" > gpurun_out/pmc2_$TAG/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc2_$TAG/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:70]
        if 'gemm' not in k: continue
        tot[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} {v / cnt[(k, c)]:16.1f}')
PY
