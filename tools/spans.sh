#!/bin/bash
# GPU box: how long a frame's encoder chain and tail chain last on their stream, alone and with other frames in flight
# (EEM_SPANS=1: HIP events inside the eager forward)
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
for s in 1 2 3 4; do
  echo "== $s frame(s) in flight (eager launches)"
  EEM_SPANS=1 python3 bench.py --steps 400 --warmup 30 --preheat 50 --cpu-seconds 0 --no-other-rows --no-side-rows --no-graph --streams $s --frames-in-flight 4 "$@" 2>&1 >/dev/null | grep EEM_SPANS | tail -$s
done
