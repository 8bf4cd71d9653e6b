#!/bin/bash
# GPU box: frame time against the length of the decoder chain: the seven decoder launches replaced by ONE sleeping wave each (no
# footprint), asleep for d us; and by 315 sleeping blocks
# the switches used here exist in the diagnostic build only: EEM_BUILD_TAG=diag EEM_EXTRA_FLAGS=-DEEM_DIAG python -m eemflow_amd.build (before gpurun)
export EEM_LIB_PATH="$(cd "$(dirname "$0")/.." && pwd)/eemflow_amd/libeemflow_hip_diag.so"
[ -f "$EEM_LIB_PATH" ] || { echo "build the diagnostic library first" >&2; exit 1; }
run() {
  EEM_SKIP_KERNELS="dec." EEM_SKIP_SPIN_US=$1 EEM_SKIP_SPIN_BLOCKS=$2 python3 bench.py --steps 400 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
}
for d in 0.1 1 2 3 4 6 8; do echo "7 x one wave asleep $d us: $(run $d 1)"; done
for d in 0.1 1 2 3; do echo "7 x 315 blocks asleep $d us: $(run $d 315)"; done
