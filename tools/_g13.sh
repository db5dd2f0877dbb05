cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
EMD_NT=7 timeout -k 10 400 python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py tests/test_d_gpu.py -q -x -m gpu > gpurun_out/r2m/t_nt7.log 2>&1; echo "t rc=$?"; tail -2 gpurun_out/r2m/t_nt7.log
run() { echo "NT=$1 XCD=$2: $(EMD_NT=$1 EMD_SEP_XCD=$2 DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; }
run 0 0 && run 0 1 && run 1 1 && run 2 1 && run 3 1 && run 4 1 && run 7 1 && run 0 0 && run 3 1 && run 3 0
