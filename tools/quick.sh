#!/bin/bash
# GPU box: the timed loop (20 and 400 steps), the single-frame latency and the per-launch table of the timed loop's configuration
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-other-rows "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f frames/s (20 steps)  %.1f (400 steps)  latency %.4f' % (d['value'], d['value_long'] or 0, d.get('latency_ms_b1') or 0))
n=d['schedule_frames_per_launch']
for k in d['kernels']: print('   %-52s %8.2f us /%d = %6.2f  blocks %d' % (k['name'], k['us'], n, k['us']/n, k['workgroups']))
rs=d.get('roofline_single_frame_launch')
if rs: print('   single-frame launches:', rs['kernels_us'])"
