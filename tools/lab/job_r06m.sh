#!/bin/bash
# r06m: (1) tail round on small tiles: parity + whole-step A/B (the library's split and tools/lab/tail_split_ab.py were removed after this session: profiles/r06m_tail_split.txt); (2) captured step (hipGraph, ABI 9): parity test, bench lines eager vs --graph at 1 / 2 / 3 / 8 clips
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "captured_step or dropout_on" > gpurun_out/r06m_pytest_graph.log 2>&1; tail -15 gpurun_out/r06m_pytest_graph.log
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "tail_round or persistent or gemm_plain or fold or dropout or sgd or head_attn or causal or embed" > gpurun_out/r06m_pytest.log 2>&1; tail -5 gpurun_out/r06m_pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:16s} {d['value']:8.1f} clips/s  {d['ms_per_step']:8.3f} ms  frac {d['roofline']['frac']:.4f}  loss {d['config']['final_loss']}  enqueue {d['host']['enqueue_ms_per_step']} ms")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
OUT=gpurun_out/r06m_graph.txt; : > $OUT
for B in 3 1 2 8; do
  for rep in 1 2; do
    timeout 600 python bench.py --batch $B --steps 40 --warmup 8 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06m_ab.json 2>gpurun_out/r06m_ab.err; line "B=$B eager" gpurun_out/r06m_ab.json >> $OUT
    timeout 600 python bench.py --graph --batch $B --steps 40 --warmup 8 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06m_ab.json 2>gpurun_out/r06m_graph_B$B.err; line "B=$B graph" gpurun_out/r06m_ab.json >> $OUT
  done
done
cat $OUT; tail -5 gpurun_out/r06m_graph_B3.err
OUT=gpurun_out/r06m_tail_split.txt; : > $OUT
for cfg in "16 30 6" "12 30 6" "20 30 6" "64 15 4" "256 10 3"; do
  set -- $cfg
  for rep in 1 2; do
    for mode in one split; do
      timeout 600 python tools/lab/tail_split_ab.py $mode --batch $1 --steps $2 --warmup $3 --no-cpu-baseline --no-also --no-gemm-trace > gpurun_out/r06m_ab.json 2>gpurun_out/r06m_ab.err; line "B=$1 $mode" gpurun_out/r06m_ab.json >> $OUT
    done
  done
done
cat $OUT
