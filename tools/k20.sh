#!/bin/bash
# GPU box: the driver's invocation (--steps 20 --warmup 5) with different pre-heat lengths, 4 repetitions each
run() { for r in 1 2 3 4; do python3 bench.py --gpus 1 --steps ${K:-20} --warmup ${W:-5} --cpu-seconds 0 --no-other-rows "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], end=' ')"; done; echo " <- K=${K:-20} W=${W:-5} $*"; }
for p in 0 40 100 200 300; do run --preheat $p; done
K=300 W=30 run --preheat 0
K=300 W=30 run --preheat 100
