#!/bin/bash
# GPU box: gconvb.hip (bf16-piece convs) against gconv16.hip on the E-RAFT and EEMFlow+ rows; tools/gconvb_ab.sh [minblk ...]
export BENCH_N=8 BENCH_WARM=3
echo "== off"; EEM_NO_GCONVB=1 python3 tools/bench_eraft.py 1; EEM_NO_GCONVB=1 python3 tools/bench_eraft.py 4; EEM_NO_GCONVB=1 python3 tools/bench_plus.py
for m in "$@"; do
  echo "== EEM_GCONVB_MINBLK=$m"
  EEM_GCONVB_MINBLK=$m python3 tools/bench_eraft.py 1; EEM_GCONVB_MINBLK=$m python3 tools/bench_eraft.py 4; EEM_GCONVB_MINBLK=$m python3 tools/bench_plus.py
done
