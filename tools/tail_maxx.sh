#!/bin/bash
# GPU box: frames/s with the decoder launches capped at <n> blocks along x (EEM_TAIL_MAXX; 0 = one pixel tile per block)
for m in "$@"; do
  echo "== EEM_TAIL_MAXX=$m"; EEM_TAIL_MAXX=$m bash tools/quick.sh
done
