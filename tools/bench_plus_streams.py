#!/usr/bin/env python3
"""EEMFlow+ with several frames in flight (one module / context per HIP stream): tools/bench_plus_streams.py [streams]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd.eemflow_plus import EEMFlow_cdc                  # noqa: E402
from eemflow_amd.plus_weights import seeded_from_shapes           # noqa: E402
from eemflow_amd.weights import synthetic_voxel_pair              # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h, w = 720, 1280
nets, streams = [], []
for _ in range(ns):
    net = EEMFlow_cdc("", 3, 5).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
    net = net.cuda()
    net.change_imagesize((h, w))
    net.frames_in_flight = ns
    nets.append(net)
    streams.append(torch.cuda.Stream())
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, 1, h, w))
torch.cuda.synchronize()
with torch.no_grad():
    for phase, n in (("warm", 2 * ns), ("timed", 10 * ns)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % ns]):
                nets[i % ns](e1, e2)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
print(f"EEMFlow+ {w}x{h} batch=1 streams={ns}: {dt*1e3:.2f} ms/forward, {1/dt:.1f} frames/s, host enqueue {(t1-t0)/n*1e3:.2f} ms/forward")
