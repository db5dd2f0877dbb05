EMD_T_WGRAD_STREAM=1 timeout -k 10 300 python bench.py --workload T --no-cpu-baseline --no-riders > gpurun_out/wgs.log 2>&1; tail -5 gpurun_out/wgs.log | cut -c1-300
