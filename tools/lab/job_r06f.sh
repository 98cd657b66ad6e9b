#!/bin/bash
# r06f: op-level + model tests on the library with direct (splitk == 1) weight-gradient accumulation; same-box A/B of the whole step against the
# previous commit's library (avt_amd/libavt_prev.so) at 3 / 16 / 64 / 256 clips per GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x  > gpurun_out/r06f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06f_pytest.log
tail -4 gpurun_out/r06f_pytest.log
OUT=gpurun_out/r06f_ab.txt; : > $OUT
for B in 3 16 64 256; do
  for rep in 1 2; do
    for lib in libavt_prev.so libavt_hip.so; do
      AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/$lib timeout 600 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06f_tmp.json 2>/dev/null
      python - $B $lib gpurun_out/r06f_tmp.json >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(f"B={sys.argv[1]:>4s} {sys.argv[2]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
    done
  done
done
cat $OUT
