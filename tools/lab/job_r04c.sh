#!/bin/bash
# r04c: full GPU suite (one-node classifier + CE, device-RNG classifier dropout at the bench head, torch-exact resize, exhaustive HSV),
# then whole-step A/B of: plain epilogue stores, nt in LayerNorm / SGD, nt attention stores, degree-6 GELU polynomial
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
for v in hip plainst lnnt attst gelu6 hip lnnt attst gelu6; do
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
for v in hip lnnt attst hip lnnt attst; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py ln attn sgd 2>&1 | grep -v amdgpu.ids
done | tee $O/kbench_stream.txt
