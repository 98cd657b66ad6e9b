#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gelu_table or epilogue" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
