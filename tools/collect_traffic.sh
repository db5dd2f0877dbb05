# Collects HBM traffic of the dominant kernels with rocprofv3 PMC counters, FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (they do not fit one pass on gfx950), and writes gpurun_out/pmc_traffic.json (copied to profiles/rNN_pmc_traffic.json) with the hash of
# the kernel sources it was measured on (bench.py reports the traffic only while that hash matches).  Graphs S and A: the step totals.
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced
# streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/K_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload K --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/D_$c -- python3 $GRAFT_REPO_ROOT/tools/dprofile.py > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/S_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload S --no-cpu-baseline --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/A_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload A --no-cpu-baseline --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
  # graphs X and T (round 4): step totals as for S and A -- bench.py --profile-clean runs the 1 + 3 steps and nothing else
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/X_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload X --no-cpu-baseline --profile-clean --steps 3 --warmup 1 > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/G_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload G --no-cpu-baseline --profile-clean --steps 3 --warmup 1 > /dev/null 2>&1
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/T_$c -- python3 $GRAFT_REPO_ROOT/bench.py --workload T --no-cpu-baseline --profile-clean --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
out = {}
# graphs S, A, X, T and G: HBM bytes of one step = everything the 4 executions (1 warm-up + 3 timed, eager) moved / 4
# (T: its hipGraph capture is off under the profiler, and the trainer's one warm-up tower of 32 x 32 px before it is noise)
for wl in ("S", "A", "X", "T", "G"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot[c] = sum(float(r["Counter_Value"]) for f in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (wl, c)) for r in csv.DictReader(open(f)))
    out[wl + ":*step total*"] = {"launches_sampled": 0, "FETCH_SIZE_KiB_avg": tot["FETCH_SIZE"] / 4, "WRITE_SIZE_KiB_avg": tot["WRITE_SIZE"] / 4,
                                 "hbm_bytes_per_step_corrected": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / 4,
                                 "hbm_bytes_per_launch_corrected": 0.0}
for wl in ("K", "D"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (wl, c)):
            for r in csv.DictReader(open(f)):
                nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if "rocclr" in nm: continue
                if wl == "K" and int(r["Grid_Size"]) < 100000: continue
                per[nm][c].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for nm, d in per.items():
        if wl == "K":  # the [32,512,512,1] launches only: bench K also times one [256,512,512,1] burst (the largest grid)
            g0 = min(g for v in d.values() for g, _ in v)
            d = {c: [x for g, x in v if g == g0] for c, v in d.items()}
        else:
            d = {c: [x for _, x in v] for c, v in d.items()}
        n = max(len(v) for v in d.values())
        f_kib = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1)
        w_kib = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1)
        out[wl + ":" + nm] = {"launches_sampled": n, "FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                              "hbm_bytes_per_launch_corrected": (2.0 * f_kib + w_kib) * 1024.0}
for v in out.values(): v["steps_sampled"] = 4   # tools/dprofile.py: 1 warm-up + 3 timed forwards; bench K: the [32,512,512,1] launches
json.dump({"csrc_sha16": bench.csrc_sha16(), "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B read requests as 64 B)", "kernels": out}, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out.items(): print(f"{k:60s} n={v['launches_sampled']:4d} fetch {v['FETCH_SIZE_KiB_avg']/1024:10.1f} MiB(raw) write {v['WRITE_SIZE_KiB_avg']/1024:10.1f} MiB -> corrected {v['hbm_bytes_per_launch_corrected']/1e6:10.1f} MB/launch")
PY
rm -rf $OUT
