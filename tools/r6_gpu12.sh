python -m pytest tests/test_gpu_bwd_ops.py -x -q -m gpu -k "ring_kernel or concatenated" 2>&1 | tail -3
python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
python tools/wgrad_bench.py 20 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
EEM_NO_WGRAD_BX3=1 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; EEM_NO_WGRAD_BX3=1 python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1
EEM_NO_WGRAD_BX3=1 python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1
EEM_WGRAD_RING=none python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1
