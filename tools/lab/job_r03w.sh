cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base varA varB base varA varB; do echo "== $v"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_$v.so KB_BATCH=256 timeout 300 python tools/kbench.py attn 2>&1 | grep vit_attn_fwd; done
AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_varB.so timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -2
