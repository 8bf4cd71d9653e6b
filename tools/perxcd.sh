#!/bin/bash
# GPU box: tuning sweeps with four frames in flight (env overrides of the persistent grid sizes / stride-2 tilings)
run() { python3 bench.py --steps 600 --warmup 40 --cpu-seconds 0 --no-other-rows --no-side-rows --streams ${NSTR:-4} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'fps', ' '.join(str(k['us']) for k in d['kernels'][:8]))"; }
run base
for v in 100 101 102; do for c in 12 16 24; do EEM_V16_32=$v EEM_ENC_PER_XCD_P32=$c run "V16_32=$v cap=$c"; done; done
for v in 100 101; do for c in 12 16 24; do EEM_V32_64=$v EEM_ENC_PER_XCD_P64=$c run "V32_64=$v cap=$c"; done; done
