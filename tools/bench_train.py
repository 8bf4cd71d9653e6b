#!/usr/bin/env python3
"""EEMFlow training-step timing on the GPU box (BASELINE configs[2]: MVSEC 346x260, batch 32; configs[3] shape b8 720p)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd import EEMFlow
from eemflow_amd.train import EEMFlowTrainer
from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair

b, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 260, 346)))
net = EEMFlow("", 5, 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
net = net.cuda().train(); net.change_imagesize((h, w))
tr = EEMFlowTrainer(net, lr=1e-4, num_steps=1000)
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).cuda() for a in synthetic_gt(2, b, h, w))
for _ in range(3): tr.step(e1, e2, gt, va)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = int(os.environ.get("EEM_BT_N", "10"))
for _ in range(n): loss, m, _ = tr.step(e1, e2, gt, va)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
flop = 3 * 2 * (0.9114e9 if (h, w) == (260, 346) else 7.2913e9 * (h * w) / (720 * 1280)) * b
print(f"EEMFlow train step {w}x{h} batch={b}: {dt*1e3:.2f} ms/step, {b/dt:.1f} samples/s, ~{flop/dt/1e12:.1f} TFLOP/s (3x fwd), loss {loss:.4f}")
