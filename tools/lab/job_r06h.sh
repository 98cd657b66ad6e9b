#!/bin/bash
# r06h: (1) judge's item 5 with counters (r06g's results were lost with the session): fabric reads per launch of each weight-gradient shape, automatic
# split factor against XCD-aligned ones; (2) this box's baseline line; (3) kernel traces of the 3- and 16-clip steps on the round-6 library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06h_wgrad_xcd.txt; : > $OUT
for cfg in "qkv 0" "qkv 8" "qkv 16" "proj 0" "proj 24" "proj 32" "fc1 0" "fc1 8" "fc2 0" "fc2 8"; do
  set -- $cfg
  timeout 300 python tools/lab/wgrad_xcd.py $1 $2 20 2>&1 | grep splitk >> $OUT
  d=gpurun_out/r06h_pmc/$1_$2
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d -o p --output-format csv -- python tools/lab/wgrad_xcd.py $1 $2 3 > gpurun_out/r06h_pmc_$1_$2.log 2>&1
  python - $d >> $OUT <<'PY'
import csv, glob, sys
v, r = [], []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] != 'FETCH_SIZE': continue
        if 'gemm_w4' in row['Kernel_Name']: v.append(float(row['Counter_Value']))
        if 'splitk_reduce' in row['Kernel_Name']: r.append(float(row['Counter_Value']))
if v: print(f'      fabric reads: gemm_w4_kernel {2 * sum(v) / len(v) * 1024 / 1e9:.3f} GB per launch ({len(v)} launches), splitk_reduce {2 * sum(r) / max(len(r), 1) * 1024 / 1e9:.3f} GB  (FETCH_SIZE in KB x 2: gfx950 correction, MI355X_MICROARCH.md HBM section)')
PY
done
rm -rf gpurun_out/r06h_pmc
cat $OUT
timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-also > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; cut -c1-600 gpurun_out/r06h_bench.json
for B in 3 16; do
  timeout 600 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06h_bench_B$B.json 2> gpurun_out/r06h_bench_B$B.err; cut -c1-400 gpurun_out/r06h_bench_B$B.json
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06h_B$B -o t --output-format csv -- python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_r06h_B$B.log 2>&1
  python tools/trace_summary.py gpurun_out/prof_r06h_B$B/t_kernel_trace.csv 10 90 > gpurun_out/r06h_kernel_trace_B$B.txt 2>&1; head -30 gpurun_out/r06h_kernel_trace_B$B.txt
  rm -rf gpurun_out/prof_r06h_B$B
done
