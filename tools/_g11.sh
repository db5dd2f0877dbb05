cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout -k 10 300 python -m pytest tests/test_split_gpu.py -q -x -m gpu -k "variants_agree" > gpurun_out/r2k/t1.log 2>&1; echo "t1 rc=$?"; tail -3 gpurun_out/r2k/t1.log
GB_VARIANTS=3,6 GB_AUX=0 GB_STAMPS=1 timeout -k 10 300 python tools/gemm_split_bench.py middle,middle_res,enc64_728_728,dec128_256_256 > gpurun_out/r2k/gb.log 2>&1; echo "gb rc=$?"; grep -v "CUs used" gpurun_out/r2k/gb.log
