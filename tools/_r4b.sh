cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gputests.log
