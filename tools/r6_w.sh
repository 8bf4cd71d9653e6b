#!/bin/bash
export EEM_BT_N=300
echo "spread:"; python3 tools/wgrad_bench.py 20 C 2>&1 | grep -v amdgpu | tail -10 | awk '{print $1,$2,$3,$4, $(NF-3), $(NF-1)}'
echo "burst:"; EEM_WGRAD_BURST=1 python3 tools/wgrad_bench.py 20 C 2>&1 | grep -v amdgpu | tail -10 | awk '{print $1,$2,$3,$4, $(NF-3), $(NF-1)}'
for i in 1 2; do echo -n "spread "; python3 tools/bench_train.py 2>/dev/null; echo -n "burst  "; EEM_WGRAD_BURST=1 python3 tools/bench_train.py 2>/dev/null; done
echo -n "spread "; python3 tools/bench_train.py 8 720 1280 2>/dev/null; echo -n "burst  "; EEM_WGRAD_BURST=1 python3 tools/bench_train.py 8 720 1280 2>/dev/null
echo -n "spread "; python3 tools/bench_eraft_train.py 2>/dev/null | tail -1; echo -n "burst  "; EEM_WGRAD_BURST=1 python3 tools/bench_eraft_train.py 2>/dev/null | tail -1
