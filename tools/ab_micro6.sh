#!/bin/bash
# GPU box: the three launch-shape changes after the end-of-round note (flow head on 32-pixel blocks, lookup blocks of 16 pixels, one-launch
# pyramid) - their tests, then same-box A/B runs through the environment switches -> gpurun_out/ab_micro6.txt
out=gpurun_out/ab_micro6.txt
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_eraft.py tests/test_gpu_plus.py tests/test_gpu_bwd_ops.py -x -q -m gpu 2>&1 | tail -5
run() { echo "--- $1"; shift; env "$@" BENCH_N=20 BENCH_WARM=3 timeout 200 python tools/bench_eraft.py $B 2>&1 | tail -1; }
for rep in 1 2; do
B=1
run "b1 default" X=0
run "b1 flow head 64-pixel form" EEM_FEWOUT_WIDE=0
run "b1 lookup 64" EEM_LOOKUP_PX=64
run "b1 lookup 32" EEM_LOOKUP_PX=32
run "b1 pool chain" EEM_POOL_CHAIN=1
run "b1 all old" EEM_FEWOUT_WIDE=0 EEM_LOOKUP_PX=64 EEM_POOL_CHAIN=1
done
B=4
run "b4 default" X=0
run "b4 lookup 32" EEM_LOOKUP_PX=32
run "b4 lookup 16" EEM_LOOKUP_PX=16
run "b4 pool chain" EEM_POOL_CHAIN=1
run "b4 default" X=0
for rep in 1 2; do
echo "--- plus default"; timeout 200 python tools/bench_plus.py 1 2>&1 | tail -1
echo "--- plus pool chain"; EEM_POOL_CHAIN=1 timeout 200 python tools/bench_plus.py 1 2>&1 | tail -1
done
} > $out 2>&1
cat $out
