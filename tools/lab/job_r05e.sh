#!/bin/bash
# round 5, session e: cache policy of the weight-gradient kernel's operand streams and slab stores (judge's item 6); which torch-native kernels are left in the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS="libavt_hip.so libavt_w4ant.so libavt_w4bnt.so libavt_w4abnt.so libavt_w4slabnt.so"
for lib in $LIBS libavt_hip.so; do
  echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 600 python tools/lab/w4_policy.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05e_w4_us.txt 2>&1
cat gpurun_out/r05e_w4_us.txt
for lib in $LIBS; do
  n=$(basename $lib .so)
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r05e_pmc/$n -o p --output-format csv -- python tools/lab/w4_policy.py 3 > gpurun_out/r05e_pmc_$n.log 2>&1
done
python - <<'PY' | tee gpurun_out/r05e_w4_fetch.txt
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r05e_pmc/*')):
    tot = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'gemm_w4' in row['Kernel_Name'] and row['Counter_Name'] == 'FETCH_SIZE':
                tot[row.get('Grid_Size', '?')].append(float(row['Counter_Value']))
    print(d.split('/')[-1], {g: round(2 * sum(v) / len(v) * 1024 / 1e9, 3) for g, v in tot.items()}, 'GB per launch (FETCH_SIZE in KB x 2, MI355X_MICROARCH.md HBM section), by grid size')
PY
for lib in $LIBS libavt_hip.so; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05e_bench_tmp.json 2>/dev/null
  python - $lib <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05e_bench_tmp.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], flush=True)
PY
done | tee gpurun_out/r05e_steps.txt
timeout 600 python tools/lab/find_native_kernels.py 32 > gpurun_out/r05e_native_kernels.txt 2>&1; tail -60 gpurun_out/r05e_native_kernels.txt
