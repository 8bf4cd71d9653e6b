python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_train.py -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_eraft.py -q -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 400 --warmup 30 --long-steps 0 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/s', d['value'], 'ms/step', d['ms_per_step'])"; done
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; EEM_WALK3_TRAIN=0 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids; EEM_WALK3_TRAIN=0 python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
python tools/bench_eraft.py 4 2>&1 | grep -v amdgpu.ids | tail -1; EEM_WALK3_ERAFT=0 python tools/bench_eraft.py 4 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_eraft.py 1 2>&1 | grep -v amdgpu.ids | tail -1; EEM_WALK3_ERAFT=0 python tools/bench_eraft.py 1 2>&1 | grep -v amdgpu.ids | tail -1
