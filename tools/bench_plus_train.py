#!/usr/bin/env python3
"""EEMFlow+ (EEMFlow_cdc) training step through the autograd route on the GPU box: forward (5 predictions) + sequence loss + backward +
torch AdamW.  usage: tools/bench_plus_train.py [batch] [h] [w]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
import torch.nn.functional as F                                   # noqa: E402
from eemflow_amd.eemflow_plus import EEMFlow_cdc                  # noqa: E402
from eemflow_amd.plus_weights import seeded_from_shapes           # noqa: E402
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair   # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = int(sys.argv[3]) if len(sys.argv) > 3 else 768
dev = "cuda:0"
net = EEMFlow_cdc("", 3, 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.to(dev).train()
net.change_imagesize((h, w))
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8)
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, b, h, w))


def step():
    opt.zero_grad()
    _, preds = net(e1, e2)
    loss = 0.0
    for i, p in enumerate(preds):                                 # coarse -> fine predictions against the resized ground truth
        g = F.interpolate(gt, size=p.shape[-2:], mode="bilinear", align_corners=False) * (p.shape[-1] / gt.shape[-1])
        loss = loss + 0.8 ** (len(preds) - 1 - i) * (p - g).abs().mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    return loss


step()
torch.cuda.synchronize()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"EEMFlow+ training step {w}x{h} b{b}: {dt * 1e3:.1f} ms/step = {b / dt:.2f} samples/s, loss {float(loss):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, host enqueue {host * 1e3:.1f} ms/step")
