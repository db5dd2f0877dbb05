cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/traceT -- python3 bench.py --workload T --steps 4 --warmup 2 --profile-clean --no-cpu-baseline > $O/benchT_trace.json 2> $O/benchT_trace.err
tail -c 300 $O/benchT_trace.json; echo
python tools/trace_overlap.py $O/traceT 0.6
rm -rf $O/traceT
