cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -q -x -m gpu > gpurun_out/r2r/t.log 2>&1; echo "t rc=$?"; tail -3 gpurun_out/r2r/t.log
for w in 0 1 0 1; do
EMD_T_WGRAD_STREAM=$w timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders --steps 5 --warmup 2 > gpurun_out/r2r/bench_T$w.json 2> gpurun_out/r2r/bench_T$w.err || { tail -5 gpurun_out/r2r/bench_T$w.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r2r/bench_T$w.json')); print('wgrad stream $w:', d['ms_per_step'], d['value'], d.get('loss_first_tower'))"
done
