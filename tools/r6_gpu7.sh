python -m pytest tests/test_gpu_bwd_ops.py -x -q -m gpu -k "few_output" 2>&1 | tail -3
python -m pytest tests/test_gpu_eraft_train.py tests/test_gpu_plus_train.py -x -q -m gpu 2>&1 | tail -3
for r in 0 1; do echo "== EEM_NO_WGRAD_FEW=$r"; EEM_NO_WGRAD_FEW=$r python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -1; done
