#!/bin/bash
# GPU box: the one-launch tail weight gradient with parts removed (EEM_TW_DBG bits: 1 atomics, 2 k-loop, 4 G staging), alone on one stream
for d in 0 1 2 3 6 7; do
EEM_TW_DBG=$d EEM_NO_WGRAD_STREAM=1 bash tools/step_timeline.sh r06tw pad4_kernel tools/bench_train.py; echo "EEM_TW_DBG=$d: $(grep 'x  1  wgrad_tail' gpurun_out/r06tw/timeline.txt)"
done
