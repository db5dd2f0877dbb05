# FETCH / WRITE of graph D's standalone depthwise launches per shape (separate PMC passes, gfx950 FETCH correction x2), next to the
# algorithmic bytes: writes gpurun_out/dw_traffic.txt (copied to profiles/r02_dw_traffic_by_shape.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dw_traffic; rm -rf $O; mkdir -p $O
python3 $R/tools/dw_traffic.py > $O/plain.json 2> $O/plain.err || { tail -3 $O/plain.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/dw_traffic.py > /dev/null 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, json
rows = json.load(open("$O/plain.json"))
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/%s/*/*counter_collection.csv" % c)[0]
    recs = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if "dw3x3" in r["Kernel_Name"]]
    recs.sort()
    assert len(recs) == 5 * len(rows), (len(recs), len(rows))
    val[c] = [sum(v for _, v in recs[5 * i + 1:5 * i + 5]) / 4 * 1024 for i in range(len(rows))]
out = ["graph D [32,512,512,1]: standalone depthwise launches per shape (tools/dw_traffic.sh; time = plain run, counters = separate PMC passes;",
       "fetched = 2 x FETCH_SIZE (gfx950 correction), bytes per launch in MB, rate = algorithmic bytes / time)",
       f"{'layer':18s} {'n':>3s} {'HxW':>5s} {'C':>4s} {'s':>2s} {'r':>3s} {'out':>7s} {'us':>8s} {'algo MB':>9s} {'fetched':>9s} {'written':>9s} {'traffic/algo':>12s} {'GB/s':>7s} {'of 8 TB/s':>9s}"]
ta = tt = tus = 0.0
for i, r in enumerate(rows):
    fe, wr = 2 * val["FETCH_SIZE"][i], val["WRITE_SIZE"][i]
    a = r["algorithmic_bytes"]
    out.append(f"{r['layer']:18s} {r['count']:3d} {r['H']:5d} {r['C']:4d} {r['stride']:2d} {r['rate']:3d} {'split32' if r['split32_out'] else 'fp32':>7s} "
               f"{r['us']:8.1f} {a/1e6:9.1f} {fe/1e6:9.1f} {wr/1e6:9.1f} {(fe+wr)/a:12.3f} {a/r['us']/1e3:7.0f} {a/r['us']/8e6:9.3f}")
    ta += a * r["count"]; tt += (fe + wr) * r["count"]; tus += r["us"] * r["count"]
out.append(f"all 49 launches: algorithmic {ta/1e9:.2f} GB, traffic {tt/1e9:.2f} GB ({tt/ta:.3f} x), {tus/1e3:.2f} ms -> {ta/tus/1e3:.0f} GB/s algorithmic = {ta/tus/8e6:.3f} of 8 TB/s")
open("$R/gpurun_out/dw_traffic.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O
