#!/bin/bash
python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo -n "default            "; python3 tools/bench_plus.py 2>/dev/null | tail -1
echo -n "MINPX_JOBS=30000   "; EEM_PLUS_WNC_MINPX_JOBS=30000 python3 tools/bench_plus.py 2>/dev/null | tail -1
done
tools/plus_timeline.sh r6_tl_default
