cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
SEB_STAMPS=1 SEB_KNOB=sep_pipe2 SEB_VALS=0,2 SEB_ONLY=deconv1_a,deconv1_dual,cnn2,deconv0_dual timeout -k 10 500 python tools/sep_epi_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/sep_stamps2.log
