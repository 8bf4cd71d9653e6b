#!/bin/bash
# GPU box: the interleaved tile walk (EEM_WALK3 layer masks): frames/s over 400 steps, parity tests under both walks, FETCH_SIZE / WRITE_SIZE per frame
# walk mode 3 on layers: ENC_1_2 = 1, ENC_2_2 = 3, ENC_2_3 = 4, ENC_3_2 = 6, ENC_3_3 = 7  (bit masks)
for m in 0 2 26 218; do
  echo "== EEM_WALK3=$m"
  EEM_WALK3=$m python bench.py --steps 400 --warmup 30 --long-steps 0 --cpu-seconds 0 --no-other-rows --no-side-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "timed_configuration or forward_many" 2>&1 | tail -2
EEM_WALK3=218 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "timed_configuration or forward_many" 2>&1 | tail -2
EEM_WALK3=26 EEM_COLWALK=0 bash tools/pmc_batch.sh r06_pmc_w3 10 EEM_WALK3=26 EEM_COLWALK=0
EEM_WALK3=218 EEM_COLWALK=0 bash tools/pmc_batch.sh r06_pmc_w3b 10 EEM_WALK3=218 EEM_COLWALK=0
