cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo "TWO_STREAMS=$v: $(EMD_D_TWO_STREAMS=$v DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; done
for v in 5 -1; do echo "TWO_STREAMS=0 variant $v: $(EMD_SPLIT_VARIANT=$v EMD_D_TWO_STREAMS=0 DP_N=8 timeout -k 10 200 python tools/dprofile.py 2>&1 | grep ms/step)"; done
