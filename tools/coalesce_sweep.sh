#!/bin/bash
# frames/s of the timed loop for coalescing width x chains in flight x grid hint (bench.py --coalesce / --streams / --frames-in-flight)
# usage (GPU box): tools/coalesce_sweep.sh > gpurun_out/coalesce_sweep.txt
cd "$(dirname "$0")/.."
run() {  # co ns hint
  local hint=""
  [ "$3" != "0" ] && hint="--frames-in-flight $3"
  python bench.py --steps 400 --warmup 40 --long-steps 0 --coalesce $1 --streams $2 $hint --cpu-seconds 0 --no-side-rows --no-other-rows 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('co $1 ns $2 hint $3 :', d['value'], 'frames/s  host enqueue ms/step', d['host_enqueue_ms_per_step'], ' dominant', d['roofline']['kernel'], d['roofline']['kernel_us'])"
}
for cfg in "1 4 0" "4 2 0" "4 2 4" "4 3 0" "4 3 2" "5 2 0" "8 2 0" "8 2 4" "8 1 1" "8 1 4" "10 2 0" "16 1 1" "16 2 0" "2 4 0" "2 3 0"; do
  run $cfg
done
