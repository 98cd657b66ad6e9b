#!/bin/bash
# r04a: cache policy of the GEMM epilogue stores (plain / sc1 write-through / nt) and of the A stream (nt) -- per-launch A/B,
# whole-step bench and fabric traffic of two launches.  Libraries: make -C avt_amd/csrc variant V=... DEFS=...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm or linear" > $O/pytest_base.log 2>&1; tail -2 $O/pytest_base.log
AVT_HIP_LIB=$L/libavt_sc1.so timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm or linear" > $O/pytest_sc1.log 2>&1; tail -2 $O/pytest_sc1.log
for v in hip sc1 ntst sc1ntA ntA hip sc1; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/lab/two_wg.py 0 2>&1 | tail -1
done | tee $O/two_wg.txt
for v in hip sc1 sc1ntA hip sc1; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py gemm 2>&1 | grep -v "^---"
done | tee $O/kbench.txt
B="--steps 10 --warmup 3 --no-cpu-baseline"
for v in hip sc1 sc1ntA ntA ntst hip sc1; do
  AVT_HIP_LIB=$L/libavt_$v.so timeout 600 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v $O/bench_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pv = d['roofline']['dominant_kernel']['per_variant_tflops']
    print(f"{sys.argv[1]:10s} {d['value']:8.1f} clips/s {d['ms_per_step']:8.2f} ms gemm {d['roofline']['dominant_kernel']['achieved']:.0f} " + ' '.join(f"{k[14:]}={v:.0f}" for k, v in pv.items() if k.startswith('gemm_8p')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done | tee $O/bench.txt
for v in hip sc1 sc1ntA; do
  for shp in "504320 3072 768" "504320 768 3072" "504320 2304 768"; do
    for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $pass | cut -d' ' -f1); d=$O/pmc_${v}_$(echo $shp | tr ' ' x)_$n
      AVT_HIP_LIB=$L/libavt_$v.so timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $d -o p --output-format csv -- python tools/one_gemm.py $shp NT 0 4 > $d.log 2>&1
      python - "$v $shp" $d <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[2] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'gemm' not in row['Kernel_Name']: continue
        tot[row['Counter_Name']] += float(row['Counter_Value']); cnt[row['Counter_Name']] += 1
print(sys.argv[1], ' '.join(f'{k} {v / cnt[k]:.0f}' for k, v in sorted(tot.items())))
PY
      rm -rf $d
    done
  done
done | tee $O/pmc.txt
