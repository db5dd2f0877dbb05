"""Dev tool: from a rocprofv3 --kernel-trace CSV, how busy was the GPU over the timed steps -- union of kernel intervals (any kernel
running), mean concurrency, idle gaps -- and per kernel name: count, total, mean.  Usage: trace_overlap.py <dir> [skip_frac]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:50], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip      # drop the warm-up part
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
for s, e, *_ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in rows)
print(f"{len(rows)} kernels over {span/1e6:.2f} ms: some kernel running {100*busy/span:.1f} % of the time, mean concurrency while busy {tot/busy:.2f}, kernel time {tot/1e6:.2f} ms")
print(f"idle gaps: {len(gaps)}, total {sum(gaps)/1e6:.2f} ms, median {sorted(gaps)[len(gaps)//2]/1e3 if gaps else 0:.1f} us")
q = collections.Counter(r[3] for r in rows)
print("queues:", dict(q))
