python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "tail_weight or dedicated or oracle_autograd" 2>&1 | tail -3
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
bash tools/step_timeline.sh r06train3 pad4_kernel tools/bench_train.py; grep "wgrad_tail" gpurun_out/r06train3/timeline.txt
