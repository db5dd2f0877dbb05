cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; rm -rf $O; mkdir -p $O
EMD_D_TWO_STREAMS=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/dprofile.py > /dev/null 2>&1 || exit 1
python3 $R/tools/trace_seq.py $O/kt 4 > $O/d_sequence_end.txt
rm -rf $O/kt
tail -45 $O/d_sequence_end.txt
