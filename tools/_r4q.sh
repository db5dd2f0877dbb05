for v in 1 0 1 0; do
  EMD_T_FUSE_PREP=$v timeout -k 10 300 python bench.py --workload A --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('A fuse_prep=$v', d['ms_per_step'])"
done
