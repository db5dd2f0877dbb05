cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload T --no-cpu-baseline --no-riders --steps 3 --warmup 1 > $O/bench_T.json 2> $O/bench_T.err || { tail -5 $O/bench_T.err; exit 1; }
python3 $R/tools/trace_groups.py $O/kt 70 > $O/groups.txt
rm -f $O/kt/*/*kernel_trace.csv
