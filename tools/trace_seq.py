"""Dev tool: the launches of the LAST repetition of a rocprofv3 --kernel-trace csv, in order: python tools/trace_seq.py DIR NREP"""
import csv, glob, sys
d, nrep = sys.argv[1], int(sys.argv[2])
import os
f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)   # the newest run (local gpurun_out/ keeps old ones)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rocclr" not in r["Kernel_Name"] and "at::native" not in r["Kernel_Name"]]
n = len(rows) // nrep
tot = 0.0
for r in rows[-n:]:
    g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{dur:9.1f} us  {tot/1e3:7.2f} ms  {r['Kernel_Name'][:60]:60s} grid {g}")
