#!/bin/bash
# GPU box: the weight-gradient calls of one E-RAFT training step (640x480, 12 iterations, batch 4), counted by shape
EEM_WGRAD_LOG=1 python tools/bench_eraft_train.py 2>&1 | grep "^WGRAD" > /tmp/wg.log
python - <<'P'
import collections
lines = open('/tmp/wg.log').read().splitlines()
c = collections.Counter(lines)
steps = 0
for l, n in c.items():
    steps = max(steps, 1)
print(f"{len(lines)} calls in the run; distinct shapes:")
for l, n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{n:5d}  {l}")
P
