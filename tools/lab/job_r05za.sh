#!/bin/bash
# round 5, session za: the small-shape tile rules (gemm.hip) -- sweep again (column 0 = the new automatic choice), small-batch steps, the GPU suite, the headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lab/small_batch_gemm_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05za_small_batch_gemm_sweep.txt; head -9 gpurun_out/r05za_small_batch_gemm_sweep.txt
python tools/lab/head_gemm_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05za_head_gemm_sweep.txt
for B in 3 8 16 32; do
  timeout 600 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/r05za_bench_B$B.json 2>/dev/null
  python - $B <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r05za_bench_B{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('B', sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['frac'], flush=True)
PY
done | tee gpurun_out/r05za_small_batch.txt
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05za_pytest.log 2>&1; tail -3 gpurun_out/r05za_pytest.log
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05za_bench_tmp.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05za_bench_tmp.json').read().strip().splitlines()[-1]); print('B 256', d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
done | tee -a gpurun_out/r05za_small_batch.txt
