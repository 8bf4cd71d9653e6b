#!/bin/bash
# GPU box: everything profiles/r06_* is made of (release + diagnostic build in the tree)
export EEM_COMMIT=${EEM_COMMIT:-unknown}
tools/profile_gpu.sh r06 10 > gpurun_out/r06_profile.log 2>&1
tools/marginal.sh > gpurun_out/r06_marginal.txt 2>&1
tools/profile_round6.sh r06 > gpurun_out/r06_round6.log 2>&1
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_gpu_tests.txt 2>&1
ls gpurun_out/r06 gpurun_out | head -80
