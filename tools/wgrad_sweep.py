#!/usr/bin/env python3
"""GPU box: one weight-gradient shape family swept over rows and widths (what a slice costs, what a k-step costs).
tools/wgrad_sweep.py cin cout kh kw stride n  h1,h2,...  w1,w2,..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.wgrad_bench import run  # noqa: E402

cin, cout, kh, kw, stride, n = (int(v) for v in sys.argv[1:7])
hs = [int(v) for v in sys.argv[7].split(",")]
ws = [int(v) for v in sys.argv[8].split(",")]
for w in ws:
    for h in hs:
        us, tf = run((f"{cin}->{cout} {kh}x{kw} s{stride} n{n} {h}x{w}", cin, cout, kh, kw, stride, n, h, w), 20)
        print(f"{cin}->{cout} {kh}x{kw} s{stride} n={n} {h:4d}x{w:4d}: {us:8.1f} us  {tf:6.1f} TFLOP/s", flush=True)
