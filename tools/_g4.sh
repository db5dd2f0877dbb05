cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 600 python -m pytest tests/test_split_gpu.py tests/test_d_gpu.py tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
tail -3 gpurun_out/r2d/pytest.log
EMD_D_TWO_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d/profD1 -- python3 tools/dprofile.py > gpurun_out/r2d/dprofile1.log 2>&1 && python tools/trace_seq.py gpurun_out/r2d/profD1 4 > gpurun_out/r2d/D_seq_single_stream.txt && python tools/prof_summary.py gpurun_out/r2d/profD1 30 > gpurun_out/r2d/D_stats_single.txt; rm -rf gpurun_out/r2d/profD1
timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" > gpurun_out/r2d/dforward.log
EMD_D_SPLIT_OUT=0 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" | sed 's/^/nosplitout: /' >> gpurun_out/r2d/dforward.log
EMD_D_DECONV_FUSED=0 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep "D forward" | sed 's/^/nodeconvfused: /' >> gpurun_out/r2d/dforward.log
cat gpurun_out/r2d/dforward.log
timeout -k 10 900 python bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc=$?"
grep "\[bench\]" gpurun_out/r2d/bench.err | tail -12
