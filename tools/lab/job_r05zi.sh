#!/bin/bash
# round 5, session zi: the LayerNorm fold from 60000 token rows on -- GPU suite, small-batch steps, the default line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r05zi_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05zi_pytest.log; tail -4 gpurun_out/r05zi_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee gpurun_out/r05zi_smoke.txt
for B in 3 8 16 24 32; do
  timeout 600 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r05zi_tmp.json 2>/dev/null
  python - $B <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r05zi_tmp.json').read().strip().splitlines()[-1]); print('B', sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['frac'], flush=True)
PY
done | tee gpurun_out/r05zi_steps.txt
timeout 900 python bench.py > gpurun_out/r05zi_bench.json 2> gpurun_out/r05zi_bench.err; cut -c1-200 gpurun_out/r05zi_bench.json
