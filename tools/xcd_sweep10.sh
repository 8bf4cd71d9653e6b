#!/bin/bash
# coalesced timed loop (400 steps): blocks per XCD of the persistent encoder kernels, chains and coalescing width
cd "$(dirname "$0")/.."
run() {
  env "$@" python bench.py --steps 400 --warmup 40 --long-steps 0 --cpu-seconds 0 --no-side-rows --no-other-rows $EXTRA 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$* $EXTRA :', d['value'])"
}
EXTRA=""
run A=0
run EEM_ENC_PER_XCD_E1=24
run EEM_ENC_PER_XCD_E1=16
run EEM_ENC_PER_XCD_F16=24
run EEM_ENC_PER_XCD_F32=24
run EEM_ENC_PER_XCD_F64=24
run EEM_ENC_PER_XCD=24
EXTRA="--streams 3"; run A=0
EXTRA="--coalesce 16 --streams 2"; run A=0
EXTRA="--coalesce 12 --streams 2"; run A=0
EXTRA="--coalesce 8 --streams 3"; run A=0
EXTRA="--coalesce 10 --streams 2 --frames-in-flight 1"; run A=0
