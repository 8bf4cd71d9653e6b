#!/usr/bin/env python3
"""Soak run of the operator route on the GPU box: 60 E-RAFT training steps (packed-weight cache, side-stream weight gradients) and
200 E-RAFT / EEMFlow+ inference forwards; device memory after warm-up / at the end - nothing may grow, the loss must fall."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd.eemflow_plus import EEMFlow_cdc                  # noqa: E402
from eemflow_amd.eraft import ERAFT                               # noqa: E402
from eemflow_amd.eraft_weights import seeded_from_shapes          # noqa: E402
from eemflow_amd import plus_weights                              # noqa: E402
from eemflow_amd.train import sequence_loss                       # noqa: E402
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair   # noqa: E402

dev = torch.device("cuda:0")


def used():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 2**20


b, h, w = 2, 256, 320
net = ERAFT("", 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.to(dev).train()
net.change_imagesize((h, w))
opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=5e-5, eps=1e-8)
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, b, h, w))
losses, mem0 = [], None
for step in range(60):
    opt.zero_grad()
    loss, _ = sequence_loss(net(e1, e2, iters=6)[1], gt, va, 0.8)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    losses.append(float(loss))
    if step == 5:
        torch.cuda.synchronize()
        mem0 = used()
torch.cuda.synchronize()
print(f"E-RAFT training: loss {losses[0]:.4f} -> {losses[-1]:.4f}, device memory {mem0:.0f} -> {used():.0f} MiB")
assert losses[-1] < losses[0] and used() - mem0 < 64

net = net.eval()
plus = EEMFlow_cdc("", 3, 5).eval()
plus.load_state_dict({k: torch.from_numpy(v) for k, v in plus_weights.seeded_from_shapes({k: tuple(v.shape) for k, v in plus.state_dict().items()}, 0).items()})
plus = plus.to(dev)
plus.change_imagesize((h, w))
with torch.no_grad():
    for i in range(200):
        net(e1, e2, iters=6)
        plus(e1, e2)
        if i == 5:
            torch.cuda.synchronize()
            mem1 = used()
torch.cuda.synchronize()
print(f"inference: device memory {mem1:.0f} -> {used():.0f} MiB")
assert used() - mem1 < 64
print("soak ok")
