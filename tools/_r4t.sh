timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "cout1 or wgrad or gradients_smooth or golden" 2>&1 | tail -2
for i in 1 2; do timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('T', d['ms_per_step'])"; done
