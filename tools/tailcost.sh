#!/bin/bash
# GPU box: is the tail's cost beside the encoders its footprint or its latency in the stream?  The ten tail launches are replaced
# by one sleeping wave each (EEM_SKIP_SPIN_US), which keeps the stream's dependency chain and its duration but occupies nothing.
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="$1" EEM_SKIP_SPIN_US="$2" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows "${@:3}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
}
T="tail head;dec.;tail up"
echo "whole frame:                 $(run "" 0 "$@")"
echo "tail skipped:                $(run "$T" 0 "$@")"
for us in 1 2.5 5 8; do echo "tail = 10 sleeping waves of $us us: $(run "$T" $us "$@")"; done
echo "encoders skipped, tail only: $(run "enc." 0 "$@")"
