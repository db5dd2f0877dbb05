cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout -k 10 600 python -m pytest tests/test_graph_exec_gpu.py -m gpu -q -x > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
tail -15 gpurun_out/r2i/pytest.log
grep -q "rc=0" gpurun_out/r2i/pytest.log || exit 1
timeout -k 10 400 python bench.py --workload D --no-cpu-baseline > gpurun_out/r2i/benchD.json 2> gpurun_out/r2i/benchD.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r2i/benchD.json')); print(d['value'], d['ms_per_step'], d.get('native_executor'))"
