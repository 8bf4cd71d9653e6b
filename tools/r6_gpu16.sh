python -m pytest tests/test_gpu_bwd_ops.py tests/test_gpu_train.py -q -m gpu -k "ring_kernel or gradients or dedicated" 2>&1 | tail -2
echo "== EEM_WGRAD_WALK=1 (default)"; python tools/wgrad_bench.py 20 C 2>&1 | grep -v amdgpu.ids
echo "== EEM_WGRAD_WALK=0"; EEM_WGRAD_WALK=0 python tools/wgrad_bench.py 20 C 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; EEM_WGRAD_WALK=0 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids; EEM_WGRAD_WALK=0 python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1; EEM_WGRAD_WALK=0 python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1
