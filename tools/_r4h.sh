cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
timeout -k 10 300 python tools/b4_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/b4_ab.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profT -- python3 bench.py --workload T --steps 3 --warmup 1 --profile-clean --no-cpu-baseline --no-graph > $O/benchT_prof.json 2> $O/benchT_prof.err
python tools/prof_summary.py $O/profT 60 > $O/benchT_stats.txt; cp $O/profT/*/*kernel_stats.csv $O/benchT_stats.csv; rm -rf $O/profT
head -70 $O/benchT_stats.txt
timeout -k 10 300 python bench.py --workload T --no-cpu-baseline > $O/benchT.json 2> $O/benchT.err; tail -c 600 $O/benchT.json
