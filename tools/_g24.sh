cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SB_STAMPS=1 timeout -k 10 300 python tools/sep_bench.py 2>&1 | grep -v amdgpu.ids
for nt in 3 7 3 7; do echo "NT=$nt: $(EMD_NT=$nt DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; done
