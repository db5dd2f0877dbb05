cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
for v in -1 5 6 4 -1 5; do
  EMD_SPLIT_VARIANT=$v DP_N=8 timeout -k 10 200 python tools/dprofile.py > gpurun_out/r2l/d_v$v.log 2>&1 || exit 1
  echo "variant $v: $(grep -i "ms" gpurun_out/r2l/d_v$v.log | tail -2 | tr '\n' ' ')"
done
