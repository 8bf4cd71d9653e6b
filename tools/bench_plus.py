#!/usr/bin/env python3
"""EEMFlow+ (EEMFlow_cdc) timing on the GPU box: 1280x720, 5 input channels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, w = 720, 1280
net = EEMFlow_cdc("", 3, 5).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.cuda(); net.change_imagesize((h, w))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, h, w))
with torch.no_grad():
    for _ in range(5): net(e1, e2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = int(os.environ.get("EEM_BP_N", "40"))
    for _ in range(n): net(e1, e2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"EEMFlow+ {w}x{h} batch={b}: {dt*1e3:.2f} ms/forward, {b/dt:.1f} frames/s, {75.3*b/dt/1e3:.1f} TFLOP/s")
