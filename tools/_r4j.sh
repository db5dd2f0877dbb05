cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
T_SPECS="wgrad_msplit=1;wgrad_msplit=2;wgrad_tile=64;wgrad_msplit=1,wgrad_tile=64;wgrad_msplit=2,wgrad_tile=64" timeout -k 10 900 python tools/t_knob_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/t_ab.log
