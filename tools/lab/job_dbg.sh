#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/lab/dbg_gelu.py 2>&1 | grep -v amdgpu.ids | tail -30
