# round 4: the never-written gradient (TO.DwGrad) -- tests, then the training step A/B (EMD_T_DW_BN_BWD 1 / 0)
timeout -k 10 600 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "never_written or affine_in or batched_per_image or streams_and_graph" > gpurun_out/dwbn_tests.log 2>&1; tail -6 gpurun_out/dwbn_tests.log
for v in 1 0 1 0; do
  EMD_T_DW_WGRAD=$v timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dw_wgrad=$v', d['ms_per_step'])"
done
