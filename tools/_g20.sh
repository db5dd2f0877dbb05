cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py -q -x -m gpu -k "dw" 2>&1 | tail -2
EMD_DW_XCD=0 bash tools/dw_traffic.sh > gpurun_out/dw_traffic_xcd0.log 2>&1; tail -17 gpurun_out/dw_traffic_xcd0.log
cp gpurun_out/dw_traffic.txt gpurun_out/dw_traffic_xcd0.txt
bash tools/dw_traffic.sh
for v in 0 1 0 1; do echo "DW_XCD=$v: $(EMD_DW_XCD=$v DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; done
