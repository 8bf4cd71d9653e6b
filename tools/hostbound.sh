#!/bin/bash
# GPU box: is the timed loop waiting for the host?  host_enqueue_ms_per_step against ms_per_step, whole frame / tail skipped / empty
run() {
  EEM_SKIP_KERNELS="$1" python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --no-side-rows "${@:2}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ms/step', d['ms_per_step'], 'host enqueue', d['host_enqueue_ms_per_step'])"
}
echo "whole frame:    $(run "")"
echo "tail skipped:   $(run "tail head;dec.;tail up")"
echo "encoder skipped:$(run "enc.")"
echo "all skipped:    $(run "enc.;tail;dec.")"
echo "whole, eager:   $(run "" --no-graph)"
