cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03l_pytest.log 2>&1; tail -4 gpurun_out/r03l_pytest.log
timeout 900 python bench.py > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03l_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['executed_frac'], d['host'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
