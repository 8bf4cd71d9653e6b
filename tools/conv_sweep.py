import os, sys, itertools
sys.path.insert(0, os.getcwd())
import torch
from eemflow_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
cases = [(4, 64, 64, 64, 80, 3), (4, 64, 96, 32, 40, 3), (4, 96, 96, 32, 40, 3), (4, 96, 128, 16, 20, 3), (4, 128, 128, 16, 20, 3),
         (4, 128, 256, 16, 20, 1), (2, 64, 64, 64, 80, 3), (2, 96, 96, 32, 40, 3), (2, 128, 256, 16, 20, 1), (4, 64, 64, 64, 80, 1),
         (2, 384, 128, 16, 20, 3), (4, 256, 128, 32, 40, 3), (4, 128, 128, 60, 80, 3), (1, 64, 64, 64, 80, 3), (8, 64, 64, 64, 80, 3)]
for (n, ci, co, h, w, k) in cases:
    x = torch.randn(n, ci, h, w, device=dev, requires_grad=True)
    conv = torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
    outs = {}
    for mode in ("0", "1"):
        os.environ["EEM_NO_GCONV16"] = mode
        x.grad = None; conv.zero_grad()
        y = ops.conv2d(conv, x, act=ops.ACT_RELU)
        g = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        y.backward(g)
        outs[mode] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone())
    e = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(outs["0"], outs["1"])]
    print((n, ci, co, h, w, k), "fwd %.2e dgrad %.2e wgrad %.2e" % tuple(e), "<<<<" if max(e) > 1e-4 else "")
