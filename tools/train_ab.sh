python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do echo "EEM_NO_WGRAD_STREAM=$v"; EEM_NO_WGRAD_STREAM=$v python3 tools/bench_train.py 2>/dev/null; EEM_NO_WGRAD_STREAM=$v python3 tools/bench_train.py 8 720 1280 2>/dev/null; done
