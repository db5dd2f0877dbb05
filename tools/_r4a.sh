cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python3 tools/gemm_traffic.py > $O/gemm_traffic_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/r4a"
for n in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
    per = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{n}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_split" in r["Kernel_Name"]:
                per[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(per.items()):
        print(n, k, "n=%d" % len(v), " ".join("%.0f" % x for x in v))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profD -- python3 bench.py --workload D --steps 10 --warmup 2 --profile-clean > $O/benchD_single.json 2> $O/benchD_single.err
python tools/prof_summary.py $O/profD 40 > $O/benchD_single_stats.txt; cat $O/benchD_single_stats.txt | head -30
rm -rf $O/profD $O/pmc_*
timeout -k 10 300 python bench.py --workload D --no-riders --no-cpu-baseline > $O/benchD.json 2> $O/benchD.err; tail -c 1500 $O/benchD.json
