#!/bin/bash
# GPU box: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) against the runtime's default, timed loop and latency
run() {
  python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ms/step', d['ms_per_step'], 'latency', d.get('latency_ms_b1'))"
}
for v in 0 1 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v: $(HIP_FORCE_DEV_KERNARG=$v run)"; done
echo "unset: $(run)"
