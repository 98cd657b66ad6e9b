cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attn or attention" > gpurun_out/r03i_pytest.log 2>&1; tail -3 gpurun_out/r03i_pytest.log
(echo "== old swizzle (lab build of the previous source)"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so KB_BATCH=256 timeout 300 python tools/kbench.py attn; echo "== new swizzle (product)"; KB_BATCH=256 timeout 300 python tools/kbench.py attn; echo "== old again"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so KB_BATCH=256 timeout 300 python tools/kbench.py attn) > gpurun_out/r03i_attn_ab.txt 2>&1
cat gpurun_out/r03i_attn_ab.txt
for b in 3 16 32 64 128; do timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03i_bench_B$b.json 2> gpurun_out/r03i_bench_B$b.err; python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/r03i_bench_B{b}.json').read().strip().splitlines()[-1])
    print(f"B={b:>3s}: {d['value']:8.1f} clips/s  {d['ms_per_step']:8.2f} ms/step  frac {d['roofline']['frac']:.4f}  host enqueue {d['host']['enqueue_ms_per_step']} ms/step  abi calls {d['host']['abi_calls_per_step']}")
except Exception as e:
    print(b, 'FAILED', e)
PY
done
