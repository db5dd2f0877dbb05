"""Dev tool: per-launch view of one kernel from a rocprofv3 --kernel-trace csv: python tools/trace_top.py DIR NAME [N]"""
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
import os
f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)   # the newest run (local gpurun_out/ keeps old ones)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if name not in r["Kernel_Name"]:
        continue
    g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[g][0] += 1
    agg[g][1] += dur
tot = sum(v[1] for v in agg.values())
print(f"{name}: {sum(v[0] for v in agg.values())} launches, {tot/1e3:.2f} ms")
for g, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"  grid {g}  calls {c:4d}  total {t/1e3:8.2f} ms  avg {t/c:8.1f} us")
