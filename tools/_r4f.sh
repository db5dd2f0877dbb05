cd $GRAFT_REPO_ROOT
for i in 1 2 3; do SD_SHAPE=2,32,64,256,256,1 python tools/sep2_debug.py 2>&1 | grep -v amdgpu.ids | grep "differs"; done
for i in 1 2; do python tools/sep2_debug.py 2>&1 | grep -v amdgpu.ids | grep "differs"; done
