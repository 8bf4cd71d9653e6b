#!/usr/bin/env python3
"""Soak run on the GPU box: 20 000 inference frames on four contexts / streams and 300 training steps (fused trainer, side-stream
weight gradients), device memory before / after each - nothing may grow, nothing may hang."""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd import EEMFlow                                   # noqa: E402
from eemflow_amd.train import EEMFlowTrainer                      # noqa: E402
from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair   # noqa: E402

dev = torch.device("cuda:0")


def used():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 2**20


nets = []
for _ in range(4):
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
    net = net.to(dev).eval()
    net.change_imagesize((720, 1280))
    net.frames_in_flight = 4
    nets.append(net)
streams = [torch.cuda.Stream() for _ in nets]
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, 720, 1280))
with torch.no_grad():
    for i in range(8):
        with torch.cuda.stream(streams[i % 4]):
            ref = nets[i % 4](e1, e2)[1][0].clone()
    torch.cuda.synchronize()
    m0, t0 = used(), time.perf_counter()
    n = 20000
    for i in range(n):
        with torch.cuda.stream(streams[i % 4]):
            out = nets[i % 4](e1, e2)[1][0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"inference: {n} frames, {n / dt:.0f} frames/s, device memory {m0:.0f} -> {used():.0f} MiB, last == first: {torch.equal(out, ref)}")
del nets
tnet = EEMFlow("", 5, 5)
tnet.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
tnet = tnet.to(dev).train()
tnet.change_imagesize((260, 346))
tr = EEMFlowTrainer(tnet, lr=1e-4, num_steps=1000)
b1, b2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 32, 260, 346))
gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, 32, 260, 346))
for _ in range(5):
    tr.step(b1, b2, gt, va)
torch.cuda.synchronize()
m0, t0 = used(), time.perf_counter()
losses = [tr.step(b1, b2, gt, va)[0] for _ in range(300)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"training: 300 steps, {dt / 300 * 1e3:.2f} ms/step, loss {losses[0]:.4f} -> {losses[-1]:.4f}, device memory {m0:.0f} -> {used():.0f} MiB")
