cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
D_SPECS="sep_pipe2:0,1,2" timeout -k 10 400 python tools/d_knob_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/d_ab.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gputests.log
