cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so
(for st in 0 auto 3 4 0 auto; do echo "== strip $st"; if [ $st = auto ]; then unset AVT_GEMM_STRIP; else export AVT_GEMM_STRIP=$st; fi; timeout 300 python tools/lab/two_wg.py 0 2>&1 | grep tile; done) > gpurun_out/r03k_strip_time.txt 2>&1
cat gpurun_out/r03k_strip_time.txt
(for st in 0 auto 4; do if [ $st = auto ]; then unset AVT_GEMM_STRIP; else export AVT_GEMM_STRIP=$st; fi
  for shape in "504320 3072 768" "504320 2304 768"; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf gpurun_out/pmc_strip; rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_strip -o p --output-format csv -- python tools/one_gemm.py $shape NT 808 > /dev/null 2>&1
    python - "$st" "$shape" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_strip/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'gemm_8p' in row['Kernel_Name']:
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
print(f'strip {sys.argv[1]:>4s} shape {sys.argv[2]}: ' + '  '.join(f'{k} {v / n[k]:.0f}' + (f' (= {v / n[k] * 2048 / 1e9:.2f} GB fabric read)' if k == 'FETCH_SIZE' else '') for k, v in tot.items()))
PY
  done; done; done; rm -rf gpurun_out/pmc_strip) > gpurun_out/r03k_strip_pmc.txt 2>&1
cat gpurun_out/r03k_strip_pmc.txt
