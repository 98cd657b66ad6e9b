#!/bin/bash
# r06z: column-strip width of the N = 3072, K = 768 GEMMs (fc1 forward, fc2 data gradient; the fc1 weight exceeds an XCD's L2): 6 column tiles per strip (product, round 3's choice
# on the one-tile kernel) against 3, 4 and no strips on today's persistent kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06z_strip_width.txt; : > $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pv = d['roofline']['dominant_kernel'].get('per_variant_tflops', {})
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}  " + ' '.join(f"{k[15:]}={v:.0f}" for k, v in pv.items() if '<9>' in k or '<11>' in k))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for rep in 1 2; do
  for lib in hip s3 s4 s12; do
    AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/r06z_ab.json 2>gpurun_out/r06z_ab.err; line "strips: $lib" gpurun_out/r06z_ab.json >> $OUT
  done
done
cat $OUT; tail -2 gpurun_out/r06z_ab.err
