#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with EEM_EXTRA_FLAGS=-DEEM_STAMPS=<C>, C = 16 / 32 / 64): where a wave of wino4_kernel
(conv_wino4.hip) spends its cycles - the stamps of the last launch with that channel count (median over blocks)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair
L = _lib.lib()
dev = torch.device("cuda", 0)
flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in seeded_state_dict(0).values()]).contiguous()
c = ctypes.c_void_p(); _lib.check(L.eemflow_create(0, ctypes.byref(c)))
_lib.check(L.eemflow_load_weights(c, flat.data_ptr(), flat.numel(), 5, 5))
H, W = 720, 1280
_lib.check(L.eemflow_set_image_size(c, H, W, None)); _lib.check(L.eemflow_use_graph(c, 0))
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, H, W))
out = torch.empty(1, 2, H, W, device=dev)
for _ in range(3):
    _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), 1, H, W, out.data_ptr(), H, W, None))
torch.cuda.synchronize()
n = 1024 * 8 * 64
buf = (ctypes.c_ulonglong * n)()
L.eemflow_debug_read_stamps4.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.eemflow_debug_read_stamps4(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 64).astype(np.int64)       # the LAST wino4 launch wrote these: pconv3_3
nb = int((s[:, 0, 0] > 0).sum())
s = s[:nb]
ks = int(((s[0, 0, 2:60] > 0).sum()) // 2)
print('tiles per block (by k-steps): ', sorted(set(((s[:, 0, 2:60] > 0).sum(axis=1) // 2).tolist())))
print("blocks", nb, "k-steps stamped", ks)
med = lambda v: int(np.median(v))
for w in (0, 7):
    a = s[:, w]
    t0 = a[:, 0]
    print(f"wave {w}: prologue {med(a[:, 1] - t0)}")
    prev = a[:, 1]
    for l in range(ks):
        b, e = a[:, 2 + 2 * l], a[:, 3 + 2 * l]
        print(f"   k-step {l:2d}: wait+barrier {med(b - prev):6d}  body {med(e - b):6d}")
        prev = e
    print(f"   output phase (+ later tiles) {med(a[:, 60] - prev)}   total {med(a[:, 60] - t0)} cycles, {med(a[:, 61])} ticks of 10 ns -> {np.median((a[:, 60] - t0) / np.maximum(a[:, 61], 1)) / 10:.2f} GHz")
print("first start -> last end", int(s[:, :, 60].max() - s[:, :, 0].min()), "cycles; start spread", int(s[:, :, 0].max() - s[:, :, 0].min()))
