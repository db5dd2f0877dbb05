cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/wgrad_pmc
rm -rf $OUT; mkdir -p $OUT
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/wgrad_pmc.py > /dev/null 2>$OUT/$tag.err || echo "failed: $c"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_wgrad_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} avg {sum(v)/len(v):14.1f}")
PY
