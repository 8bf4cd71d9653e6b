#!/bin/bash
# GPU box, diagnostic build: the fused first-two-layers launch with phases switched off (EEM_E12_DBG): where a tile's time goes
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
for d in 0 1 2 10 14 30; do
  echo -n "EEM_E12_DBG=$d: "
  EEM_FUSE12=1 EEM_E12_DBG=$d python3 bench.py --steps 20 --warmup 5 --long-steps 0 --cpu-seconds 0 --no-other-rows --no-side-rows "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels'][0]; print(k['name'][:24], k['us'], 'us per launch of', d['schedule_frames_per_launch'], 'frames;', d['value'], 'frames/s')"
done
