#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ac; mkdir -p $O
for lib in hip lnpf2 hip lnpf2; do echo "== $lib"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$lib.so timeout 200 python tools/lab/ln_check.py 2>&1 | grep -v amdgpu.ids; done | tee $O/ln.txt
