cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-middle}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  GB_REPS=3 timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_g -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py $SHAPE > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_g/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_conv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"  {k:28s} {sum(v)/len(v):16.0f}")
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_g
done
