cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py tests/test_surface.py tests/test_multi_rank.py -m gpu -x -q > $O/train_tests.log 2>&1; echo "train tests rc=$?"; tail -5 $O/train_tests.log
T_SPECS="env:EMD_T_BN_SMALL=0" timeout -k 10 600 python tools/t_knob_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/t_ab.log
