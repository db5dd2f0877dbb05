# round 4: the never-written gradient (TO.DwGrad) -- tests, then the training step A/B (EMD_T_DW_BN_BWD 1 / 0)
true
for v in 1 0 1 0; do
  EMD_T_LAZY_RES=$v timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lazy_res=$v', d['ms_per_step'])"
done
