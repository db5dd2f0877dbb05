cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  GB_REPS=3 timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_g -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py middle > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_g/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_conv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"  {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_g
done
