#!/bin/bash
# GPU box: the tail's ten launches replaced by sleeping blocks of 576 threads (no memory traffic): is it the slots they hold?
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="$1" EEM_SKIP_SPIN_US="$2" EEM_SKIP_SPIN_BLOCKS="$3" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
}
T="tail head;dec.;tail up"
echo "whole frame:   $(run "" 0 1)"
echo "tail skipped:  $(run "$T" 0 1)"
for blocks in 1 64 315 1024; do for us in 1 3; do echo "tail = 10 launches of $blocks sleeping blocks, $us us: $(run "$T" $us $blocks)"; done; done
