cd /tmp && export TMPDIR=/tmp
first=1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_k -- python3 $GRAFT_REPO_ROOT/tools/ktune.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_k/*/*counter_collection.csv"):
    rd=csv.DictReader(open(f))
    pass
    for r in rd:
        if "k3_tile" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X","0")))>=500000: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"  {k:32s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
  first=0
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_k
done
