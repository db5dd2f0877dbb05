cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sep_pipe_gpu.py -m gpu -x -q > $O/sep.log 2>&1; echo "sep tests rc=$?"; tail -15 $O/sep.log
