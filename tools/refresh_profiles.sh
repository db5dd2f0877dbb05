# Refreshes the committed evidence of a round on the GPU box: PMC traffic, rocprofv3 kernel statistics of the bench command,
# the bench line itself.  Usage (inside gpurun): bash tools/refresh_profiles.sh r02
R=${1:-r04}
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
rm -rf gpurun_out/profAll
# graph D, the primary line, twice: SINGLE stream (full-batch launches, one after the other: avg us x launches of a family reproduces the
# line's depthwise_frac / pointwise_frac by hand) and the TWO-stream form the timed step runs (half batches overlapped); graphs X and T once
for cfg in "D_single:--workload D --steps 10 --warmup 2 --profile-clean --profile-streams single" \
           "D_two:--workload D --steps 10 --warmup 2 --profile-clean --profile-streams two" \
           "X:--workload X --steps 3 --warmup 1 --profile-clean --no-cpu-baseline" \
           "T:--workload T --steps 3 --warmup 1 --profile-clean --no-cpu-baseline --no-graph"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py $args > gpurun_out/refresh/bench${tag}_prof.json 2> gpurun_out/refresh/bench${tag}_prof.err
  if [ "${tag#D_}" != "$tag" ]; then name=${R}_benchD_kernel_stats_${tag#D_}; else name=${R}_bench${tag}_kernel_stats; fi
  python tools/prof_summary.py gpurun_out/prof_$tag 40 > gpurun_out/refresh/$name.txt
  cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/refresh/$name.csv
  rm -rf gpurun_out/prof_$tag
done
echo "profiles done"
timeout -k 10 900 python bench.py > gpurun_out/refresh/${R}_bench_all.json 2> gpurun_out/refresh/bench_all.err; echo "bench rc=$?"
grep "\[bench\]" gpurun_out/refresh/bench_all.err | tail -3
tail -c 400 gpurun_out/refresh/${R}_bench_all.json
