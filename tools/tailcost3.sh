#!/bin/bash
# GPU box: sleeping blocks with and without 16 KB of straight-line code in front (instruction-cache footprint beside the encoders)
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="$1" EEM_SKIP_SPIN_US="$2" EEM_SKIP_SPIN_BLOCKS="$3" EEM_SKIP_SPIN_CODE="$4" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
}
T="tail head;dec.;tail up"
echo "whole frame:   $(run "" 0 1 0)"
echo "tail skipped:  $(run "$T" 0 1 0)"
for code in 0 1; do for blocks in 64 315; do echo "tail = 10 launches of $blocks blocks asleep 3 us, code=$code: $(run "$T" 3 $blocks $code)"; done; done
