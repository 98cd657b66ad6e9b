#!/bin/bash
# round 5, session zd: the 3-deep ring for the head's forward layout at tiny row counts -- GPU suite, small-batch steps, headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05zd_pytest.log 2>&1; tail -3 gpurun_out/r05zd_pytest.log
for B in 3 3 8 8; do
  timeout 600 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r05zd_tmp.json 2>/dev/null
  python - $B <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05zd_tmp.json').read().strip().splitlines()[-1]); print('B', sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('final_loss'), flush=True)
PY
done | tee gpurun_out/r05zd_steps.txt
timeout 900 python bench.py --no-cpu-baseline --no-also --no-gemm-trace --steps 15 > gpurun_out/r05zd_tmp.json 2>/dev/null
python - <<'PY' | tee -a gpurun_out/r05zd_steps.txt
import json
d = json.loads(open('gpurun_out/r05zd_tmp.json').read().strip().splitlines()[-1]); print('B 256', d['value'], d['ms_per_step'], d['config'].get('final_loss'), flush=True)
PY
