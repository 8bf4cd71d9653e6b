#!/bin/bash
# GPU box: the timed loop, the single-frame latency and the per-launch table in one line each
python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f frames/s %.4f ms  latency %.4f' % (d['value'], d['ms_per_step'], d.get('latency_ms_b1') or 0))
print('   enc ', ' '.join('%.2f' % k['us'] for k in d['kernels'] if k['name'].startswith('enc.')))
print('   tail', ' '.join('%.2f' % k['us'] for k in d['kernels'] if not k['name'].startswith('enc.')))"
