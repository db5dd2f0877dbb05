"""Dev tool: summarise a rocprofv3 --kernel-trace --stats output directory (top kernels by total time)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over all calls")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{nm[:64]:64s} calls {int(r["Calls"]):5d} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} {float(r["Percentage"]):5.1f}%')
