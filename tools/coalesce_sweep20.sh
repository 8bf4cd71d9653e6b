#!/bin/bash
# the driver's invocation (--steps 20 --warmup 5) for coalescing width x chains: value (20 steps) and value_long (400 steps), three runs each
cd "$(dirname "$0")/.."
run() {  # co ns hint
  local hint=""
  [ "$3" != "0" ] && hint="--frames-in-flight $3"
  for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --coalesce $1 --streams $2 $hint --cpu-seconds 0 --no-side-rows --no-other-rows 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('co $1 ns $2 hint $3 : value(20)', d['value'], ' value_long(400)', d['value_long'])"
  done
}
for cfg in "1 4 0" "4 2 0" "4 3 0" "5 2 0" "5 4 0" "10 2 0" "8 2 0" "20 1 1" "10 1 1"; do
  run $cfg
done
