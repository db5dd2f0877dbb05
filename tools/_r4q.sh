timeout -k 10 600 python -m pytest tests/test_gan_train_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do timeout -k 10 300 python bench.py --workload A --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('A', d['ms_per_step'])"; done
