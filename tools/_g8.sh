cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_train_ops_gpu.py tests/test_gan_train_gpu.py -m gpu -q -x -s > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
grep -E "batched per-image|gradient accumulation|passed|failed|rc=" gpurun_out/r2h/pytest.log | tail -6
grep -q "rc=0" gpurun_out/r2h/pytest.log || exit 1
for mode in streams batched; do
  timeout -k 10 300 python bench.py --workload T --tower-mode $mode --no-cpu-baseline --steps 5 --warmup 2 2> gpurun_out/r2h/benchT_$mode.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['ms_per_step'])"
done
