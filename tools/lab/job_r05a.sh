#!/bin/bash
# round 5, session a: persistent-GEMM range stealing (parity under CU masks, A/B against the round-4 scheduler), new DDP tests, baseline bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/r05a_box.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "persistent or gemm" > gpurun_out/r05a_pytest_gemm.log 2>&1; tail -5 gpurun_out/r05a_pytest_gemm.log
timeout 900 python -m pytest tests/test_ddp_gpu.py -m gpu -q -x > gpurun_out/r05a_pytest_ddp.log 2>&1; tail -5 gpurun_out/r05a_pytest_ddp.log
for lib in libavt_base.so libavt_hip.so libavt_base.so libavt_hip.so; do
  echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib KB_BATCH=256 timeout 600 python tools/kbench.py gemm 2>&1 | grep -v "^---" 
done > gpurun_out/r05a_kbench.txt 2>&1
cat gpurun_out/r05a_kbench.txt
timeout 900 python bench.py --no-cpu-baseline --no-also > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; tail -c 600 gpurun_out/r05a_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['dominant_kernel']['frac'], r['dominant_kernel']['share_of_step_time'], r.get('worst_large_gemm_row', {}).get('kernel'), r.get('worst_large_gemm_row', {}).get('frac'))
print(r['dominant_kernel']['per_variant_tflops'])
PY
AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_base.so timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r05a_bench_base.json 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r05a_bench_new.json 2>/dev/null
python - <<'PY'
import json
for f in ('r05a_bench_base', 'r05a_bench_new'):
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
