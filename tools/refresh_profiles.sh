set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh
bash tools/collect_traffic.sh > gpurun_out/refresh/collect.log 2>&1 || true
cp gpurun_out/pmc_traffic.json profiles/r01_pmc_traffic.json 2>/dev/null || true
echo "traffic done"
rm -rf gpurun_out/profAll gpurun_out/profD gpurun_out/profK
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profAll -- python3 bench.py --no-cpu-baseline > gpurun_out/refresh/bench_all_prof.json 2> gpurun_out/refresh/bench_all_prof.err
python tools/prof_summary.py gpurun_out/profAll 30 > gpurun_out/refresh/all_stats.txt
cp gpurun_out/profAll/*/*kernel_stats.csv gpurun_out/refresh/all_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profD -- python3 bench.py --workload D --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/refresh/benchD_prof.json 2> gpurun_out/refresh/benchD_prof.err
python tools/prof_summary.py gpurun_out/profD 20 > gpurun_out/refresh/D_stats.txt
cp gpurun_out/profD/*/*kernel_stats.csv gpurun_out/refresh/D_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profK -- python3 bench.py --workload K --no-cpu-baseline > gpurun_out/refresh/benchK_prof.json 2> gpurun_out/refresh/benchK_prof.err
python tools/prof_summary.py gpurun_out/profK 6 > gpurun_out/refresh/K_stats.txt
cp gpurun_out/profK/*/*kernel_stats.csv gpurun_out/refresh/K_kernel_stats.csv
echo "profiles done"
timeout -k 10 700 python bench.py > gpurun_out/refresh/bench_all.json 2> gpurun_out/refresh/bench_all.err
tail -c 600 gpurun_out/refresh/bench_all.json
