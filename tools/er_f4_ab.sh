#!/bin/bash
# GPU box: E-RAFT with the encoder's 64 -> 64 convs on the Winograd F(4x4,3x3) kernel (default) against gconv16 (EEM_ERAFT_NO_F4=1)
for b in 1 4; do for v in 1 0 1 0; do echo "batch $b EEM_ERAFT_NO_F4=$v: $(EEM_ERAFT_NO_F4=$v python3 tools/bench_eraft.py $b 2>/dev/null | tail -1 | cut -c1-90)"; done; done
