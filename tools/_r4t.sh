for v in 0 1 0 1; do
  EMD_T_STAGGER=$v timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('stagger=$v', d['ms_per_step'])"
done
