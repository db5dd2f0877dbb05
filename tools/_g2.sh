cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout -k 10 900 python -m pytest tests/test_input_gpu.py tests/test_x_graph.py tests/test_d_gpu.py "tests/test_ops_gpu.py" tests/test_split_gpu.py tests/test_train_gpu.py -m gpu -q -x -s > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
