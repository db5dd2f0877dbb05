cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_k_gpu.py -q -x -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do EMD_K_XCD=$v timeout -k 10 200 python bench.py --workload K --no-cpu-baseline --no-riders --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K_XCD=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
