#!/bin/bash
# First multi-GPU session on a node with N MI355X: the scaling curve and the knobs that decide it, one table.
#   usage: tools/scale_sweep.sh [NMAX=8] [STEPS=20]
# For every N in 1 2 4 .. NMAX: bench.py as the driver launches it; at NMAX additionally the exchange variants (reduce-scatter + all-gather,
# bf16 wire, 32 / 128 MiB buckets) and RCCL channel limits (each channel is a workgroup that takes a CU from the one-workgroup-per-CU GEMM grids).
# Columns: clips/s, ms/step, exposed exchange (ms the optimizer waited after backward), GEMM family ms/step with the collectives in flight | paused.
NMAX=${1:-8}; STEPS=${2:-20}
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/scale
run() { tag=$1; shift; python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-also "$@" > gpurun_out/scale/$tag.json 2> gpurun_out/scale/$tag.err
  python - $tag gpurun_out/scale/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    c = d.get('comm', {})
    ex = max((r.get('comm_exposed_ms', 0) for r in c.get('per_rank', []) if r), default=0)
    g = c.get('gemm_family_ms_per_step', {})
    print(f"{sys.argv[1]:28s} N={d['n_gpus']} {d['value']:9.1f} clips/s {d['ms_per_step']:8.2f} ms  exposed {ex:6.2f} ms  gemm {g.get('with_collectives_in_flight', float('nan')):7.2f} | {g.get('exchange_paused', float('nan')):7.2f} ms  per-rank {min(d['per_rank_clips_per_s']):.0f}-{max(d['per_rank_clips_per_s']):.0f}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
N=1; while [ $N -le $NMAX ]; do run n$N --gpus $N; N=$((N * 2)); done
if [ $NMAX -gt 1 ]; then
  run n${NMAX}_rs_ag --gpus $NMAX --reduce-mode rs_ag
  run n${NMAX}_bf16wire --gpus $NMAX --wire-dtype bf16
  run n${NMAX}_bucket32 --gpus $NMAX --bucket-mb 32
  run n${NMAX}_bucket128 --gpus $NMAX --bucket-mb 128
  for ch in 4 8 16; do run n${NMAX}_maxch$ch --gpus $NMAX --nccl-max-nchannels $ch; done
fi
