#!/usr/bin/env python3
"""GPU box, stamps build (EEM_BUILD_TAG=wncst EEM_EXTRA_FLAGS=-DEEM_WNC_STAMPS python -m eemflow_amd.build; EEM_LIB_PATH=.../libeemflow_hip_wncst.so):
where a wave of the Winograd kernel (conv_wnc.hip) spends its cycles, for the last launch of an EEMFlow+ forward at 1280x720 with
EEM_WNC_STAMPS_JOBS jobs x EEM_WNC_STAMPS_CHUNKS chunks (compile-time; default 3 x 3: the level-2 decoder's first conv): python3 tools/wnc_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eemflow_amd import _lib
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair

h, w = 720, 1280
net = EEMFlow_cdc("", 3, 5).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.cuda(); net.change_imagesize((h, w))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, 1, h, w))
with torch.no_grad():
    for _ in range(3): net(e1, e2)
    torch.cuda.synchronize()
L = _lib.lib()
n = 256 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
assert L.eemflow_debug_read_wnc_stamps(buf, n) == 0
a = np.array(buf[:], dtype=np.int64).reshape(256, 8, 8).astype(np.float64)
a = a[a[:, :, 7].sum(axis=1) > 0]
names = ["top wait", "top barrier", "requests", "first half", "mid wait", "second half", "exchange + epilogue", "total"]
print("selected wnc launch: %d blocks with stamps; cycles per wave, mean over waves and blocks (share of total)" % len(a))
tot = a[:, :, 7].mean()
for i, nm in enumerate(names):
    print("  %-22s %9.0f  %5.1f %%" % (nm, a[:, :, i].mean(), 100 * a[:, :, i].mean() / tot))
