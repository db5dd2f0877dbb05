cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r2s/t.log 2>&1; echo "t rc=$?"; tail -5 gpurun_out/r2s/t.log
