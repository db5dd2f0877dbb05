# Dev tool: HBM traffic of the training step (bench.py --workload T, eager, 1 + 3 steps) per KERNEL NAME -- rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE in separate passes, gfx950 correction as in tools/collect_traffic.sh -- beside each kernel's launch count, so that the
# passes worth removing are picked from measurements.  Writes gpurun_out/t_traffic.txt.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/t_traffic
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/T_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload T --no-cpu-baseline --profile-clean --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
done
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/t_traffic.txt
import csv, glob, collections
per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/T_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            per[nm][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                per[nm]["n"] += 1
rows = sorted(((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / 4, 2.0 * v["FETCH_SIZE"] * 1024.0 / 4, v["WRITE_SIZE"] * 1024.0 / 4, v["n"] / 4, k) for k, v in per.items())
tot = sum(r[0] for r in rows)
print(f"training step, HBM bytes per step by kernel (4 eager steps / 4): total {tot / 1e9:.1f} GB")
print(f"{'GB/step':>9s} {'read':>8s} {'write':>8s} {'launches':>9s}  kernel")
for b, rd, wr, n, k in reversed(rows):
    if b > 0.05e9:
        print(f"{b / 1e9:9.2f} {rd / 1e9:8.2f} {wr / 1e9:8.2f} {n:9.0f}  {k[:120]}")
PY
rm -rf $OUT
cat $GRAFT_REPO_ROOT/gpurun_out/t_traffic.txt
