set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bwd_ops.py -x -q -m gpu -k "ring or infinity" 2>&1 | tail -15 > gpurun_out/t1.txt
timeout 600 python tools/wgrad_bench.py 20 > gpurun_out/wgrad_bench.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -x -q -m gpu -k "train or timed_configuration or infinity or batched_decoder" 2>&1 | tail -15 > gpurun_out/t2.txt
timeout 300 python tools/bench_train.py > gpurun_out/train.txt 2>&1; timeout 300 python tools/bench_train.py 8 720 1280 >> gpurun_out/train.txt 2>&1
EEM_NO_WGRAD_RING=1 timeout 300 python tools/bench_train.py > gpurun_out/train_old.txt 2>&1; EEM_NO_WGRAD_RING=1 timeout 300 python tools/bench_train.py 8 720 1280 >> gpurun_out/train_old.txt 2>&1
cat gpurun_out/t1.txt gpurun_out/wgrad_bench.txt gpurun_out/t2.txt gpurun_out/train.txt gpurun_out/train_old.txt
