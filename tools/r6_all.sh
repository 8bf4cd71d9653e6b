#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_now.json 2> gpurun_out/r6_bench_now.err
tail -c 300 gpurun_out/r6_bench_now.json
