#!/bin/bash
# GPU box: effect of the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) on the multi-stream rows
for q in ${QUEUES:-16 32}; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q"
for ns in ${STREAMS:-3 4 5 6 8}; do
python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-other-rows --streams $ns 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $ns', d['value'], 'fps', d['ms_per_step'], d['latency_ms_b1'], d['latency_and_pipeline'].get('fresh_buffers_frames_per_s'), d['latency_and_pipeline'].get('pipeline_frames_per_s'))"
done
done
