timeout -k 10 300 python -m pytest tests/test_train_ops_gpu.py -x -q -m gpu -k "wgrad or weight_grad" 2>&1 | tail -2
timeout -k 10 200 python tools/small_gemm_bench.py 2>&1 | grep wgrad
