#!/bin/bash
python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
EEM_PLUS_WNC_MINPX=0 python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
EEM_PLUS_WNC_MINPX=0 EEM_WNC_SMALL_MAXPX=0 python -m pytest tests/test_gpu_plus.py -m gpu -x -q 2>&1 | tail -2
