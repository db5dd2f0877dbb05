cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_d -- python3 $GRAFT_REPO_ROOT/tools/dprofile.py 2>&1 | grep -E "D forward|rror"
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_d
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_d/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//4
for r in rows[-n:]:
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if d>350 or "sep_fused" in nm: print(f'{nm[:40]:40s} grid {r["Grid_Size_X"]:>9s}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {d:9.1f} us')
PY
