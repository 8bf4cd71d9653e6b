#!/usr/bin/env python3
"""E-RAFT training step on the GPU box: forward (autograd route) + sequence loss + backward + torch AdamW, BASELINE configs[4] shape
by default (640x480, 12 iterations, batch 4).  usage: tools/bench_eraft_train.py [batch] [iters] [h] [w]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd.eraft import ERAFT
from eemflow_amd.eraft_weights import seeded_from_shapes
from eemflow_amd.train import sequence_loss
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
h = int(sys.argv[3]) if len(sys.argv) > 3 else 480
w = int(sys.argv[4]) if len(sys.argv) > 4 else 640
dev = "cuda:0"
net = ERAFT("", 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.to(dev).train()
net.change_imagesize((h, w))
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8)
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, b, h, w))
gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, b, h, w))
def step():
    opt.zero_grad()
    _, preds = net(e1, e2, iters=iters)
    loss, _ = sequence_loss(preds, gt, va, 0.8)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    return loss
step(); torch.cuda.synchronize()
if os.environ.get("ERAFT_TRAIN_GRAPH", "0") == "1":
    # the whole step as one HIP graph (torch.cuda.graph: forward, loss, backward on the autograd thread, clip, AdamW): static inputs,
    # gradients zeroed in place, a capturable optimizer, no host read inside
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8, capturable=True)
    def gstep():
        opt.zero_grad(set_to_none=False)
        _, preds = net(e1, e2, iters=iters)
        loss, _ = sequence_loss(preds, gt, va, 0.8, metrics=False)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        return loss
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            gstep()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):       # the stream of the warm-up: the library keeps packed weights per stream
        static_loss = gstep()
    torch.cuda.synchronize()
    def step():
        g.replay()
        return static_loss
    step(); torch.cuda.synchronize()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"E-RAFT training step {w}x{h} b{b} {iters} iterations: {dt * 1e3:.1f} ms/step = {b / dt:.2f} samples/s, loss {float(loss):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, host enqueue {host * 1e3:.1f} ms/step")
