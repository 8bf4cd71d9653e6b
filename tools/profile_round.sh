#!/bin/bash
# GPU box: everything profiles/r<NN>_* is made of, in one call: tools/profile_round.sh <tag>
tag=${1:-r05}
export EEM_COMMIT=${EEM_COMMIT:-unknown}
tools/profile_gpu.sh $tag 10 > gpurun_out/${tag}_profile.log 2>&1
tools/marginal.sh > gpurun_out/${tag}_marginal.txt 2>&1
tools/profile_rows.sh ${tag}_rows > gpurun_out/${tag}_rows.txt 2>&1
tools/prof_eraft_train.sh ${tag}_ertrain > gpurun_out/${tag}_ertrain.txt 2>&1
tools/pmc_e12.sh ${tag}_pmc_e12 > gpurun_out/${tag}_pmc_e12.txt 2>&1
tools/micro/dma_overlap > gpurun_out/${tag}_dma_overlap.txt 2>&1
tools/micro/dma_pieces > gpurun_out/${tag}_dma_pieces.txt 2>&1
python3 tools/copyrate.py >> gpurun_out/${tag}_dma_pieces.txt 2>/dev/null
python3 tools/voxel_many_bench.py > gpurun_out/${tag}_voxel_many.txt 2>/dev/null
tools/wgrad_nomfma.sh > gpurun_out/${tag}_wgrad_nomfma.txt 2>&1
tools/e12_phases.sh > gpurun_out/${tag}_e12_phases.txt 2>&1
ls gpurun_out/$tag
