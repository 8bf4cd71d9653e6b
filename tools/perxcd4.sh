#!/bin/bash
# GPU box: blocks per XCD of the persistent encoder kernels with four frames in flight (EEM_ENC_PER_XCD_<tag>: E1 pconv1_1, F16 / F32 / F64
# the F(4x4) kernels; 0 = the launcher's own choice)
run() { python3 bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'fps', ' '.join(str(k['us']) for k in d['kernels'][:8]))"; }
run base; run base
for c in 12 16 24 32; do EEM_ENC_PER_XCD_E1=$c run "E1=$c"; done
for c in 10 30; do EEM_ENC_PER_XCD_F16=$c run "F16=$c"; done
for c in 8 10; do EEM_ENC_PER_XCD_F32=$c run "F32=$c"; done
for c in 4 6; do EEM_ENC_PER_XCD_F64=$c run "F64=$c"; done
