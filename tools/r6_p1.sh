#!/bin/bash
python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
EEM_PLUS_WNC_MINPX=0 python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
EEM_PLUS_WNC_MINPX=0 EEM_WNC_SMALL_MAXPX=0 python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo -n "default            "; python3 tools/bench_plus.py 2>/dev/null | tail -1
done
tools/plus_timeline.sh r6_tl_default
