cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "sep_gemm" > gpurun_out/r2f/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> gpurun_out/r2f/pytest_ops.log
tail -2 gpurun_out/r2f/pytest_ops.log
timeout -k 10 400 python -m pytest tests/test_d_gpu.py -m gpu -q -x > gpurun_out/r2f/pytest_d.log 2>&1; echo "pytest d rc=$?" >> gpurun_out/r2f/pytest_d.log
tail -3 gpurun_out/r2f/pytest_d.log
grep -q "rc=0" gpurun_out/r2f/pytest_d.log || exit 1
for cfg in "" "EMD_D_PIPELINE=0" "EMD_D_SEPGEMM=1" "EMD_D_SEPGEMM=1 EMD_D_PIPELINE=0"; do
  env $cfg DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" | sed "s/^/[$cfg] /" >> gpurun_out/r2f/dforward.log
done
cat gpurun_out/r2f/dforward.log
