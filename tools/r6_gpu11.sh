python -m pytest tests/test_gpu_plus.py tests/test_gpu_plus_train.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_eraft.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_plus.py 2>&1 | grep -v amdgpu.ids; EEM_PLUS_NO_FUSE=1 python tools/bench_plus.py 2>&1 | grep -v amdgpu.ids
python tools/bench_eraft.py 2>&1 | grep -v amdgpu.ids | tail -3
