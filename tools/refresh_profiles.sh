# Refreshes the committed evidence of a round on the GPU box: PMC traffic, rocprofv3 kernel statistics of the bench command,
# the bench line itself.  Usage (inside gpurun): bash tools/refresh_profiles.sh r02
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh
bash tools/collect_traffic.sh > gpurun_out/refresh/collect.log 2>&1 || true
cp gpurun_out/pmc_traffic.json gpurun_out/refresh/${R}_pmc_traffic.json 2>/dev/null || true
cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic.json 2>/dev/null || true
echo "traffic done"
rm -rf gpurun_out/profAll
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profAll -- python3 bench.py --no-cpu-baseline > gpurun_out/refresh/bench_all_prof.json 2> gpurun_out/refresh/bench_all_prof.err
python tools/prof_summary.py gpurun_out/profAll 40 > gpurun_out/refresh/${R}_bench_all_kernel_stats.txt
cp gpurun_out/profAll/*/*kernel_stats.csv gpurun_out/refresh/${R}_bench_all_kernel_stats.csv
rm -rf gpurun_out/profAll gpurun_out/profD
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profD -- python3 bench.py --workload D --steps 10 --warmup 2 --profile-clean > gpurun_out/refresh/benchD_prof.json 2> gpurun_out/refresh/benchD_prof.err
python tools/prof_summary.py gpurun_out/profD 30 > gpurun_out/refresh/${R}_benchD_kernel_stats.txt
cp gpurun_out/profD/*/*kernel_stats.csv gpurun_out/refresh/${R}_benchD_kernel_stats.csv
rm -rf gpurun_out/profD
echo "profiles done"
timeout -k 10 900 python bench.py > gpurun_out/refresh/${R}_bench_all.json 2> gpurun_out/refresh/bench_all.err; echo "bench rc=$?"
grep "\[bench\]" gpurun_out/refresh/bench_all.err | tail -3
tail -c 400 gpurun_out/refresh/${R}_bench_all.json
