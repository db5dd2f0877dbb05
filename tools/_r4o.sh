cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_T2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_T2 -- python3 bench.py --workload T --steps 3 --warmup 1 --profile-clean --no-cpu-baseline --no-graph > gpurun_out/prof_T2.json 2> gpurun_out/prof_T2.err
python tools/prof_summary.py gpurun_out/prof_T2 45 > gpurun_out/prof_T2_stats.txt
rm -rf gpurun_out/prof_T2
cat gpurun_out/prof_T2_stats.txt
