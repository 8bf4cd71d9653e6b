#!/bin/bash
# GPU box: the default bench line and the 20-step line of the sources in the tree
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_head.json 2> gpurun_out/r06_bench_head.err
tail -c 600 gpurun_out/r06_bench_head.json
