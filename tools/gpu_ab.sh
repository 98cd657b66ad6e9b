#!/bin/bash
# A/B session: a few parity tests, then bench lines for the product library and for other builds of it.  usage: tools/gpu_ab.sh TAG [lib.so ...]
TAG=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "layernorm or cls_query or decode or embed_pos or checkpoint or g3b or config2 or cls_only" > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err; python - "$name" gpurun_out/${TAG}_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pv = d['roofline']['dominant_kernel']['per_variant_tflops']
    print(f"{sys.argv[1]:14s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.2f} ms  frac {d['roofline']['frac']:.4f}  gemm {d['roofline']['dominant_kernel']['achieved']:.0f}  8p: " + ' '.join(f"{k[14:]}={v:.0f}" for k, v in pv.items() if k.startswith('gemm_8p')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
# the product library, then every extra library given on the command line (e.g. a lab build with a variant compiled in):
#   tools/gpu_ab.sh TAG avt_amd/libavt_hip_lab.so avt_amd/libavt_variant.so
run product python bench.py $B
shift
for lib in "$@"; do
  AVT_HIP_LIB=$GRAFT_REPO_ROOT/$lib run $(basename $lib .so) python bench.py $B
done
run product_again python bench.py $B
