cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2j/bench_rank1.json 2> gpurun_out/r2j/bench_rank1.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r2j/bench_rank1.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], list(k for k in d if k.startswith('workload_')), d['workload_T'].get('allreduce'), d['workload_T']['ms_per_step'])"
tail -3 gpurun_out/r2j/bench_rank1.err
