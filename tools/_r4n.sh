# round 4: training-step fusions -- tests of the touched paths, then an A/B of one environment switch (T_AB_VAR, values 1 / 0)
timeout -k 10 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "${T_AB_TESTS:-stats or never_written or affine_in or batched_per_image}" > gpurun_out/t_ab_tests.log 2>&1; tail -4 gpurun_out/t_ab_tests.log
for v in 1 0 1 0; do
  env ${T_AB_VAR:-EMD_T_FUSE_STATS}=$v timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('${T_AB_VAR:-EMD_T_FUSE_STATS}=$v', d['ms_per_step'])"
done
