cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_graph_exec_gpu.py tests/test_abi.py -q -x 2>&1 | tail -2
timeout -k 10 300 python tools/native_bench.py 2>&1 | grep -v amdgpu
