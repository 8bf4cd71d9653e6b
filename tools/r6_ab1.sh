#!/bin/bash
export EEM_BT_N=300
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd.py tests/test_gpu_configs.py tests/test_gpu_dp_processes.py -m gpu -q 2>&1 | grep -v "^$" | tail -4
for i in 1 2 3; do python3 tools/bench_train.py 2>/dev/null; done
python3 tools/bench_train.py 8 720 1280 2>/dev/null
