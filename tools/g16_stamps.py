#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with EEM_EXTRA_FLAGS=-DEEM_G16_STAMPS): where a wave of the LDS-tiled conv spends its cycles.
tools/g16_stamps.py n cin cout h w kh kw   (default: E-RAFT's 384 -> 128 1x5 GRU conv at 60x80, batch 1)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib, ops

n, ci, co, h, w, kh, kw = (int(v) for v in (sys.argv[1:8] if len(sys.argv) > 7 else (1, 384, 128, 60, 80, 1, 5)))
dev = "cuda:0"
x = torch.randn(n, ci, h, w, device=dev)
conv = torch.nn.Conv2d(ci, co, (kh, kw), padding=(kh // 2, kw // 2)).to(dev)
with torch.no_grad():
    for _ in range(20):
        ops.conv2d(conv, x, act=ops.ACT_RELU)
torch.cuda.synchronize()
L = _lib.lib()
N = 1024 * 4 * 8
buf = (ctypes.c_ulonglong * N)()
L.eemflow_debug_read_g16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.eemflow_debug_read_g16_stamps(buf, N) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 8).astype(np.int64)
nb = int((s[:, 0, 0] > 0).sum())
s = s[:nb]
print(f"conv {ci}->{co} {kh}x{kw} {h}x{w} n={n}: {nb} blocks")
for wv in (0, 3):
    a = s[:, wv]
    med = lambda v: int(np.median(v))
    print(f"wave {wv}: prologue {med(a[:, 1] - a[:, 0])}  wait-vmcnt {med(a[:, 2])}  barrier {med(a[:, 3])}  requests {med(a[:, 4])}  compute {med(a[:, 5])}"
          f"  loop {med(a[:, 6] - a[:, 1])}  epilogue {med(a[:, 7] - a[:, 6])}  total {med(a[:, 7] - a[:, 0])} (max {int((a[:, 7] - a[:, 0]).max())})")
t0 = s[:, :, 0].min()
print("first start -> last end:", int(s[:, :, 7].max() - t0), "cycles (s_memtime);  start spread", int(s[:, :, 0].max() - t0))
