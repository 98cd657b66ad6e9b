#!/bin/bash
# r06a: the whole-model oracle tests on both routes (margins printed), the bench-size property test with the repaired past-logits assertion, a baseline bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "route or oracle or bench_size or fold" > gpurun_out/r06a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06a_pytest.log
grep -A3 "^MARGINS" gpurun_out/r06a_pytest.log > gpurun_out/r06a_margins.txt
tail -5 gpurun_out/r06a_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; cut -c1-600 gpurun_out/r06a_bench.json
