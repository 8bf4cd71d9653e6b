#!/bin/bash
python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -3
echo "MINPX=0 (every level through the Winograd kernel)"
EEM_PLUS_WNC_MINPX=0 python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
echo -n "default   "; python3 tools/bench_plus.py 2>/dev/null | tail -1
echo -n "NO_WNC=1  "; EEM_NO_WNC=1 python3 tools/bench_plus.py 2>/dev/null | tail -1
echo -n "MINPX=10000  "; EEM_PLUS_WNC_MINPX=10000 python3 tools/bench_plus.py 2>/dev/null | tail -1
done
