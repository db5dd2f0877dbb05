cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -q -x -m gpu > gpurun_out/r2q/t.log 2>&1; echo "t rc=$?"; tail -3 gpurun_out/r2q/t.log
timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders --steps 5 --warmup 2 > gpurun_out/r2q/bench_T.json 2> gpurun_out/r2q/bench_T.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r2q/bench_T.json')); print(d['ms_per_step'], d['value'])"
