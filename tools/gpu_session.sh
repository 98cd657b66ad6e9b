#!/bin/bash
# One gpurun session: GPU parity tests, the bench lines for the BASELINE configs, a kernel trace and the whole-step PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only).  usage: tools/gpu_session.sh TAG [tests|notests] [pmc|nopmc]
TAG=${1:-r02a}; TESTS=${2:-tests}; PMC=${3:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$TESTS" = tests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
  tail -15 gpurun_out/${TAG}_pytest.log
fi
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json
timeout 600 python bench.py --frames 15 --batch 128 --steps 8 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/${TAG}_bench_T15.json 2> gpurun_out/${TAG}_bench_T15.err; cut -c1-400 gpurun_out/${TAG}_bench_T15.json
timeout 600 python bench.py --model vit_large_patch16_224 --batch 96 --steps 8 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/${TAG}_bench_vitl.json 2> gpurun_out/${TAG}_bench_vitl.err; cut -c1-400 gpurun_out/${TAG}_bench_vitl.json
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} -o ${TAG} --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/prof_${TAG}.log 2>&1
python tools/trace_summary.py gpurun_out/prof_${TAG}/${TAG}_kernel_trace.csv 5 60 > gpurun_out/${TAG}_trace_summary.txt 2>&1; head -40 gpurun_out/${TAG}_trace_summary.txt
rm -f gpurun_out/prof_${TAG}/${TAG}_kernel_trace.csv
if [ "$PMC" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${TAG}/$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-trace --no-also > gpurun_out/pmc_${TAG}_$c.log 2>&1
  done
  python tools/pmc_step_summary.py gpurun_out/pmc_${TAG} 3 > gpurun_out/${TAG}_pmc_step.txt 2>&1; cat gpurun_out/${TAG}_pmc_step.txt | head -40
  rm -rf gpurun_out/pmc_${TAG}/*/*counter_collection.csv gpurun_out/pmc_${TAG}/*/*kernel_trace.csv
fi
