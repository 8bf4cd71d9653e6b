python -m pytest tests/test_gpu_bwd_ops.py -x -q -m gpu -k "ring" 2>&1 | tail -3
python tools/wgrad_bench.py 20 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; python tools/bench_train.py 8 720 1280 2>&1 | grep -v amdgpu.ids
export EEM_LIB_PATH=$PWD/eemflow_amd/libeemflow_hip_diag.so
for d in 2 6; do echo "== EEM_WG_DBG=$d"; EEM_WG_DBG=$d python tools/wgrad_bench.py 20 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7}'; done
