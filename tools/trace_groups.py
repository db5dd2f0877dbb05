"""Dev tool: group a rocprofv3 kernel trace by (kernel, grid): calls, total and average duration."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    g[(nm, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
print(f"total {tot/1e3:.2f} ms")
flt = sys.argv[3] if len(sys.argv) > 3 else ""
for (nm, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    if flt and flt not in nm: continue
    print(f"{nm[:52]:52s} grid {grid:18s} n={len(v):4d} total {sum(v)/1e3:8.2f} ms avg {sum(v)/len(v):8.1f} us")
