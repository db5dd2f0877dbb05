cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout -k 10 300 python -m pytest tests/test_split_gpu.py -q -x -m gpu -k "variants_agree or direct_epilogue" > gpurun_out/r2o/t1.log 2>&1; echo "t1 rc=$?"; tail -5 gpurun_out/r2o/t1.log
GB_VARIANTS=3,7 GB_AUX=0 GB_STAMPS=1 timeout -k 10 300 python tools/gemm_split_bench.py middle,middle_res,enc64_728_728,dec128_256_256,enc256_128_128 > gpurun_out/r2o/gb.log 2>&1; echo "gb rc=$?"; grep -v "CUs used" gpurun_out/r2o/gb.log
EMD_NT=4 GB_VARIANTS=3,7 GB_AUX=0 timeout -k 10 300 python tools/gemm_split_bench.py middle,middle_res,enc64_728_728 > gpurun_out/r2o/gb_nt.log 2>&1; echo "gb rc=$?"; grep -v "CUs used" gpurun_out/r2o/gb_nt.log
for v in -1 7 -1 7; do echo "variant $v: $(EMD_SPLIT_VARIANT=$v DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; done
