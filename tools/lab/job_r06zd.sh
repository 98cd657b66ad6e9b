#!/bin/bash
# r06zd: smoke + the full GPU suite on the round's HEAD
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06zd_smoke.log 2>&1; tail -2 gpurun_out/r06zd_smoke.log
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06zd_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06zd_pytest.log; tail -6 gpurun_out/r06zd_pytest.log
