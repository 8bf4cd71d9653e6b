#!/bin/bash
# GPU box: the persistent bf16-piece stride-1 kernels (conv_bx3p.hip) against the Winograd ones: tools/bx3p_ab.sh [masks...]
for m in "$@"; do
  echo "== EEM_BX3P=$m"
  EEM_BX3P=$m bash tools/quick.sh
done
