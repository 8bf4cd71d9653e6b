#!/bin/bash
# GPU box: tuning sweeps with four frames in flight (env overrides of the persistent grid sizes / stride-2 tilings)
run() { python3 bench.py --steps 600 --warmup 40 --cpu-seconds 0 --no-other-rows --streams ${NSTR:-4} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'fps', d['latency_ms_b1'], ' '.join(str(k['us']) for k in d['kernels'][:8]))"; }
run base; run base
for v in 1 2 3; do EEM_V16_32=$v run "V16_32=$v"; done
for v in 1 2; do EEM_V32_64=$v run "V32_64=$v"; done
EEM_ENC_PER_XCD_W64=29 run "W64=29"
EEM_ENC_PER_XCD_E1=16 run "E1=16"
EEM_ENC_PER_XCD_E1=24 run "E1=24"
EEM_ENC_PER_XCD_W16=20 run "W16=20"
EEM_ENC_PER_XCD_W16=28 run "W16=28"
EEM_ENC_PER_XCD_W32=24 run "W32=24"
EEM_ENC_PER_XCD_W32=20 run "W32=20"
