cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/parts_ab.py 2>&1 | grep -v amdgpu.ids
