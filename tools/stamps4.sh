#!/bin/bash
# GPU box: rebuild conv_wino4.hip with stamps for C = $1 (16 / 32 / 64) and print the phase table
touch eemflow_amd/csrc/conv_wino4.hip
EEM_EXTRA_FLAGS=-DEEM_STAMPS=$1 python3 -c "from eemflow_amd.build import build_library; build_library(verbose=False)"
python3 tools/stamps4.py
