cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attn or attention" > gpurun_out/r03s_pytest.log 2>&1; tail -3 gpurun_out/r03s_pytest.log
(echo "== round-3 start (lab build of the earlier source: builtin transposing reads, fenced barriers)"; AVT_HIP_LIB=$GRAFT_REPO_ROOT/avt_amd/libavt_hip_lab.so KB_BATCH=256 timeout 300 python tools/kbench.py attn; echo "== new (product)"; KB_BATCH=256 timeout 300 python tools/kbench.py attn) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03s_attn.txt
cat gpurun_out/r03s_attn.txt
