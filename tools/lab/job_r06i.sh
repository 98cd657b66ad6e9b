#!/bin/bash
# r06i: column-major walk of wide weight-gradient tile grids (libavt_hip.so) against the previous library; XCD-aligned split factors in the whole step;
# fabric reads per launch of fc1 / fc2 again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06i_wgrad_walk.txt; : > $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "accum or wgrad or weight_grad or reproducible" > gpurun_out/r06i_pytest.log 2>&1; tail -3 gpurun_out/r06i_pytest.log
for cfg in "fc1 0" "fc2 0" "qkv 0"; do
  set -- $cfg
  timeout 300 python tools/lab/wgrad_xcd.py $1 $2 20 2>&1 | grep splitk >> $OUT
  d=gpurun_out/r06i_pmc/$1_$2
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -o p --output-format csv -- python tools/lab/wgrad_xcd.py $1 $2 3 > gpurun_out/r06i_pmc_$1_$2.log 2>&1
  python - $d >> $OUT <<'PY'
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] == 'FETCH_SIZE' and 'gemm_w4' in row['Kernel_Name']: v.append(float(row['Counter_Value']))
if v: print(f'      fabric reads: gemm_w4_kernel {2 * sum(v) / len(v) * 1024 / 1e9:.3f} GB per launch ({len(v)} launches)')
PY
done
rm -rf gpurun_out/r06i_pmc
B="--steps 12 --warmup 3 --no-cpu-baseline --no-also --no-gemm-trace"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:22s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for rep in 1 2; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_prev.so timeout 600 python bench.py $B > gpurun_out/r06i_a.json 2>/dev/null; line "prev (row-major)" gpurun_out/r06i_a.json >> $OUT
  timeout 600 python tools/lab/wgrad_align_ab.py auto $B > gpurun_out/r06i_b.json 2>/dev/null; line "new walk, auto splits" gpurun_out/r06i_b.json >> $OUT
  timeout 600 python tools/lab/wgrad_align_ab.py align $B > gpurun_out/r06i_c.json 2>/dev/null; line "new walk, aligned" gpurun_out/r06i_c.json >> $OUT
done
cat $OUT
