python -m pytest tests/test_gpu_bwd_ops.py -x -q -m gpu -k "ring or concatenated" 2>&1 | tail -3
python -m pytest tests/test_gpu_eraft_train.py tests/test_gpu_plus_train.py tests/test_gpu_autograd.py -x -q -m gpu 2>&1 | tail -3
python tools/wgrad_bench.py 30 "5x1" 2>&1 | grep -v amdgpu.ids
for r in default none all; do
  if [ $r = default ]; then unset EEM_WGRAD_RING; else export EEM_WGRAD_RING=$r; fi
  echo "== EEM_WGRAD_RING=$r"; python tools/bench_eraft_train.py 2>&1 | grep -v amdgpu.ids | tail -2
done
