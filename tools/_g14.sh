cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; rm -rf $O; mkdir -p $O
for nt in 0 7; do
  EMD_NT=$nt timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$nt -- python3 $R/tools/dprofile.py > /dev/null 2>&1 || exit 1
  echo "== NT=$nt"; python3 $R/tools/prof_summary.py $O/kt$nt 16
  EMD_NT=$nt timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$nt -- python3 $R/tools/dprofile.py > /dev/null 2>&1 || exit 1
  EMD_NT=$nt timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$nt -- python3 $R/tools/dprofile.py > /dev/null 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
for c, d in (("FETCH", "$O/f$nt"), ("WRITE", "$O/w$nt")):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            per[nm + " g" + r["Grid_Size"]].append(float(r["Counter_Value"]))
    tot = 0
    for nm, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        tot += sum(v)
    print(c, "total KiB per forward", tot / 4, " -> GB", tot / 4 * 1024 / 1e9 * (2 if c == "FETCH" else 1))
    for nm, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f"   {nm[:70]:70s} n={len(v):3d} avg {sum(v)/len(v)/1024:9.1f} MiB raw")
PY
done
rm -rf $O/f0 $O/f7 $O/w0 $O/w7
