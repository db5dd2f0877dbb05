cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -4 gpurun_out/r2c/pytest.log
for b in 4 8 16 32; do DP_B=$b DP_N=5 timeout -k 10 120 python tools/dprofile.py 2>&1 | grep "D forward" >> gpurun_out/r2c/batch_sweep.log; done
EMD_D_DUAL=0 DP_N=5 timeout -k 10 120 python tools/dprofile.py 2>&1 | grep "D forward" | sed 's/^/nodual: /' >> gpurun_out/r2c/batch_sweep.log
cat gpurun_out/r2c/batch_sweep.log
timeout -k 10 900 python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r2c/bench.json
