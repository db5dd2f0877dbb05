cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sep_pipe_gpu.py -m gpu -x -q > $O/sep.log 2>&1; echo "sep tests rc=$?"; tail -3 $O/sep.log
SEB_STAMPS=1 SEB_KNOB=sep_pipe2 SEB_VALS=0,2 timeout -k 10 500 python tools/sep_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/sep_stamps.log
