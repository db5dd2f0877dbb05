for g in 4 2 8 4 2; do
  EMD_T_GROUPS=$g timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('groups=$g', d['ms_per_step'])"
done
