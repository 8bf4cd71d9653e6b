#!/bin/bash
# GPU box: headline loop with 3..8 frames in flight
for s in "$@"; do
  python3 bench.py --steps 400 --warmup 40 --cpu-seconds 0 --no-other-rows --no-side-rows --streams $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s:', d['value'], 'frames/s', d['ms_per_step'], 'ms/frame')"
done
