#!/bin/bash
# r04b: (1) full GPU suite on the product library (classifier + CE as one node, device-RNG classifier dropout, ABI 5);
# (2) store / stream policy variants, whole-step A/B; (3) per-kernel LN / SGD / attention with nt streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
L=$GRAFT_REPO_ROOT/avt_amd
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
for v in hip ntst ntst_ntP ntst_ntA ntst_snt hip ntst ntst_snt; do
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
for v in hip ntst_snt hip ntst_snt; do
  echo "== $v"; AVT_HIP_LIB=$L/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py ln attn sgd 2>&1 | grep -v amdgpu.ids
done | tee $O/kbench_stream.txt
