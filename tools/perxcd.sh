#!/bin/bash
# GPU box: persistent-grid size sweep for the encoder kernels (4 frames in flight)
run() { python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --streams ${NSTR:-4} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'fps', d['latency_ms_b1'], ' '.join(str(k['us']) for k in d['kernels'][:8]))"; }
export EEM_ENC_PER_XCD_E1=20 EEM_ENC_PER_XCD_W16=24 EEM_ENC_PER_XCD_W32=29
run base
for v in 100 101 102; do EEM_V16_32=$v run "V16_32=$v"; done
for v in 100 101; do EEM_V32_64=$v run "V32_64=$v"; done
for c in 32 48 64; do EEM_V16_32=100 EEM_ENC_PER_XCD_P32=$c run "V16_32=100 cap=$c"; done
for c in 32 48 64; do EEM_V32_64=100 EEM_ENC_PER_XCD_P64=$c run "V32_64=100 cap=$c"; done
