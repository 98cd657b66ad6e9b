#!/bin/bash
# round 5, final check: the whole GPU suite and the default bench line on the round's last commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r05zz_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05zz_pytest.log; tail -4 gpurun_out/r05zz_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee gpurun_out/r05zz_smoke.txt
timeout 900 python bench.py > gpurun_out/r05zz_bench.json 2> gpurun_out/r05zz_bench.err; cut -c1-300 gpurun_out/r05zz_bench.json
