#!/bin/bash
# r06n: the round's library on a fresh box: full GPU suite, smoke, default bench line (+ also), T = 15 / ViT-L lines, kernel trace, whole-step PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06n_smoke.log 2>&1; tail -2 gpurun_out/r06n_smoke.log
bash tools/gpu_session.sh r06n tests pmc
