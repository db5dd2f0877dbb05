cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_d -- python3 $GRAFT_REPO_ROOT/tools/dprofile.py 2>&1 | grep -E "D forward|rror"
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_d
