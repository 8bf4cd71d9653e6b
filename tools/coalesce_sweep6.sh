#!/bin/bash
# GPU box: frames per forward_many call x chains in flight, 400 steps, round 6 sources
cd "$(dirname "$0")/.."
run() {
  python bench.py --steps 400 --warmup 40 --long-steps 0 --coalesce $1 --streams $2 --cpu-seconds 0 --no-side-rows --no-other-rows 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('co $1 ns $2 :', d['value'], 'frames/s')"
}
for cfg in "10 2" "8 2" "12 2" "16 2" "10 3" "8 3" "6 3" "16 1" "10 2"; do run $cfg; done
