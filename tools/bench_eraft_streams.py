#!/usr/bin/env python3
"""E-RAFT with several frames in flight (one module / context per HIP stream): tools/bench_eraft_streams.py [batch] [streams]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd.eraft import ERAFT                               # noqa: E402
from eemflow_amd.eraft_weights import seeded_from_shapes          # noqa: E402
from eemflow_amd.weights import synthetic_voxel_pair              # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h, w, iters = 480, 640, 12
nets, streams = [], []
for _ in range(ns):
    net = ERAFT("", 5).eval()
    sd = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda()
    net.change_imagesize((h, w))
    net.frames_in_flight = ns
    nets.append(net)
    streams.append(torch.cuda.Stream())
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, h, w))
torch.cuda.synchronize()


def frame(i):
    k = i % ns
    with torch.cuda.stream(streams[k]):
        return nets[k](e1, e2, iters=iters)


with torch.no_grad():
    for i in range(2 * ns):
        frame(i)
    torch.cuda.synchronize()
    n = 6 * ns
    t0 = time.perf_counter()
    for i in range(n):
        frame(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"E-RAFT {w}x{h} iters={iters} batch={b} streams={ns}: {dt*1e3:.2f} ms/forward, {b/dt:.2f} frames/s, host enqueue {(t1-t0)/n*1e3:.2f} ms/forward")
