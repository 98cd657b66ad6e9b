cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03p_pytest.log 2>&1; tail -4 gpurun_out/r03p_pytest.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03p_bench.json 2> gpurun_out/r03p_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03p_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['executed_frac'], d['roofline']['dominant_kernel']['achieved'], d['roofline']['dominant_kernel']['per_variant_tflops'])
PY
