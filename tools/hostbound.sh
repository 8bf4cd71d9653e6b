#!/bin/bash
# GPU box: is the timed loop waiting for the host?  host_enqueue_ms_per_step against ms_per_step, whole frame / tail skipped / empty
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="$1" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows --long-steps 0 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ms/step', d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'])"
}
echo "whole frame:    $(run "")"
echo "tail skipped:   $(run "tail head;dec.;tail up")"
echo "encoder skipped:$(run "enc.")"
echo "all skipped:    $(run "enc.;tail;dec.")"
echo "whole, eager:   $(run "" --no-graph)"
