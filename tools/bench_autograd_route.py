#!/usr/bin/env python3
"""GPU box: the reference's own training statements (train_mvsec.py:241-258) on eemflow_amd.EEMFlow - model(e1, e2) -> sequence_loss ->
loss.backward() -> clip -> torch AdamW - against the fused trainer (tools/bench_train.py).  usage: bench_autograd_route.py [batch h w]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd import EEMFlow                                   # noqa: E402
from eemflow_amd.train import sequence_loss                       # noqa: E402
from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair   # noqa: E402

b, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 260, 346)
dev = torch.device("cuda:0")
net = EEMFlow("", 5, 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
net = net.to(dev).train()
net.change_imagesize((h, w))
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8)
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, b, h, w))


def step():
    opt.zero_grad()
    _, preds = net(e1, e2)
    loss, _ = sequence_loss(preds, gt, va, 0.8)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"EEMFlow autograd route {w}x{h} b{b}: {dt * 1e3:.2f} ms/step = {b / dt:.0f} samples/s, host enqueue {host * 1e3:.2f} ms/step, loss {float(loss):.4f}")
