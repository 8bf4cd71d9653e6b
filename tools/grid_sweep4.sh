#!/bin/bash
# GPU box: frames/s (four frames in flight, rotated inputs) against the blocks per XCD of the persistent encoder kernels (EEM_ENC_PER_XCD_<tag>)
run() { env "$@" python3 bench.py --steps 400 --warmup 50 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
echo "default            $(run A=1)  $(run A=1)"
for v in 8 10 12 16 20 24 32; do echo "E1=$v (pconv1_1)   $(run EEM_ENC_PER_XCD_E1=$v)"; done
for v in 8 10 12 15 20 30; do echo "F16=$v (pconv1_2)  $(run EEM_ENC_PER_XCD_F16=$v)"; done
for v in 8 10 12 15; do echo "F32=$v (pconv2_x)  $(run EEM_ENC_PER_XCD_F32=$v)"; done
for v in 4 6 8; do echo "F64=$v (pconv3_x)  $(run EEM_ENC_PER_XCD_F64=$v)"; done
echo "default            $(run A=1)  $(run A=1)"
