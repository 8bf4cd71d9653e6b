python -m pytest tests/test_gpu_train.py tests/test_gpu_plus_train.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
EEM_TRAIN_LATE_STATS=1 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids
EEM_UPBWD_THREADS=1 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids
bash tools/step_timeline.sh r06train2 pad4_kernel tools/bench_train.py; head -95 gpurun_out/r06train2/timeline.txt
