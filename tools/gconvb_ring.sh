#!/bin/bash
# GPU box, diagnostic build: E-RAFT 640x480 x 12 at batch 1 and 4 with three weight slots in gconvb.hip's tiles of up to six rows (EEM_GCONVB_RING=3)
# against the two-slot ring
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
for b in 1 4; do
  for r in 2 3 2 3; do
    echo -n "batch $b ring $r: "; EEM_GCONVB_RING=$r BENCH_N=10 BENCH_WARM=3 python3 tools/bench_eraft.py $b 2>/dev/null | tail -1 | cut -c1-90
  done
done
