#!/bin/bash
# GPU box: E-RAFT 640x480 x 12 at batch 1 and 4 with the two-slot weight ring everywhere (EEM_GCONVB_RING=2) against the default
for b in 1 4; do
  for r in 2 3 2 3; do
    echo -n "batch $b ring $r: "; EEM_GCONVB_RING=$r BENCH_N=10 BENCH_WARM=3 python3 tools/bench_eraft.py $b 2>/dev/null | tail -1 | cut -c1-90
  done
done
