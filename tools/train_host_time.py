#!/usr/bin/env python3
"""GPU box: host time of the calls of one EEMFlow training step (C3: 346x260 batch 32): forward_backward, optimizer_step, stats_wait."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd import EEMFlow, _lib
from eemflow_amd.train import EEMFlowTrainer
from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair

b, h, w = 32, 260, 346
net = EEMFlow("", 5, 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
net = net.cuda().train(); net.change_imagesize((h, w))
tr = EEMFlowTrainer(net, lr=1e-4, num_steps=1000)
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).cuda() for a in synthetic_gt(2, b, h, w))
for _ in range(5): tr.step(e1, e2, gt, va)
L = _lib.lib()
names = ["eemflow_forward_backward", "eemflow_train_stats_async", "eemflow_optimizer_step", "eemflow_train_stats_wait"]
acc = {n: 0.0 for n in names}
class Timed:
    def __init__(self, f, n): self.f, self.n = f, n
    def __call__(self, *a):
        t = time.perf_counter(); r = self.f(*a); acc[self.n] += time.perf_counter() - t; return r
class Proxy:
    def __getattr__(self, n):
        f = getattr(L, n)
        return Timed(f, n) if n in acc else f
_lib_lib = _lib.lib
_lib.lib = lambda: Proxy()
torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
for _ in range(n): tr.step(e1, e2, gt, va)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("step %.3f ms; host per step: %s; python rest %.3f ms" % (dt / n * 1e3, {k: round(v / n * 1e3, 3) for k, v in acc.items()}, (dt - sum(acc.values())) / n * 1e3))
