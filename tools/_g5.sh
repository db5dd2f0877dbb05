cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "sep_gemm or sep_dual" > gpurun_out/r2e/pytest_ops.log 2>&1; echo "pytest ops rc=$?" >> gpurun_out/r2e/pytest_ops.log
tail -3 gpurun_out/r2e/pytest_ops.log
grep -q "rc=0" gpurun_out/r2e/pytest_ops.log || exit 1
timeout -k 10 400 python -m pytest tests/test_d_gpu.py tests/test_train_gpu.py -m gpu -q -x -s > gpurun_out/r2e/pytest_d.log 2>&1; echo "pytest d rc=$?" >> gpurun_out/r2e/pytest_d.log
tail -3 gpurun_out/r2e/pytest_d.log
timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" > gpurun_out/r2e/dforward.log
EMD_D_SEPGEMM=0 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" | sed 's/^/nosepgemm: /' >> gpurun_out/r2e/dforward.log
EMD_D_TWO_STREAMS_FUSED=1 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" | sed 's/^/sepgemm+twostreams: /' >> gpurun_out/r2e/dforward.log
cat gpurun_out/r2e/dforward.log
EMD_D_TWO_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/profD1 -- python3 tools/dprofile.py > gpurun_out/r2e/dprofile1.log 2>&1 && python tools/trace_seq.py gpurun_out/r2e/profD1 4 > gpurun_out/r2e/D_seq_single_stream.txt && python tools/prof_summary.py gpurun_out/r2e/profD1 30 > gpurun_out/r2e/D_stats_single.txt; rm -rf gpurun_out/r2e/profD1
head -12 gpurun_out/r2e/D_stats_single.txt
