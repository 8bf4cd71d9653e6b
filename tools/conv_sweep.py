"""GPU box: operator-level conv (ops.Conv2d) with each dedicated kernel against the generic one of the same library:
EEM_NO_GCONV16 (forward / data gradient) and EEM_NO_WGRAD_WIDE (weight gradient), over E-RAFT's layer shapes; and against torch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd import ops                                       # noqa: E402

torch.manual_seed(0)
dev = "cuda:0"
cases = [(4, 64, 64, 64, 80, (3, 3)), (4, 96, 96, 32, 40, (3, 3)), (4, 128, 128, 16, 20, (3, 3)), (4, 128, 256, 16, 20, (1, 1)),
         (4, 384, 128, 60, 80, (1, 5)), (4, 384, 128, 60, 80, (5, 1)), (4, 324, 256, 60, 80, (1, 1)), (4, 256, 192, 60, 80, (3, 3)),
         (4, 192, 126, 60, 80, (3, 3)), (4, 128, 256, 60, 80, (3, 3)), (4, 256, 576, 60, 80, (1, 1)), (2, 128, 128, 46, 62, (3, 3)),
         (1, 64, 64, 30, 44, (3, 3)), (4, 64, 96, 64, 80, (3, 3)), (4, 256, 2, 60, 80, (3, 3)), (3, 48, 80, 33, 52, (1, 5))]
worst = 0.0
for (n, ci, co, h, w, k) in cases:
    x = torch.randn(n, ci, h, w, device=dev, requires_grad=True)
    conv = torch.nn.Conv2d(ci, co, k, padding=(k[0] // 2, k[1] // 2)).to(dev)
    g = torch.randn(n, co, h, w, device=dev)
    outs = {}
    for mode in ("fast", "generic"):
        os.environ["EEM_NO_GCONV16"] = os.environ["EEM_NO_WGRAD_WIDE"] = "1" if mode == "generic" else "0"
        x.grad = None
        conv.zero_grad()
        y = ops.conv2d(conv, x, act=ops.ACT_RELU)
        y.backward(g)
        outs[mode] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x.grad = None
    conv.zero_grad()
    yr = torch.relu(conv(x))
    yr.backward(g)
    ref = (yr.detach(), x.grad, conv.weight.grad, conv.bias.grad)
    e = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(outs["fast"], outs["generic"])]
    r = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(outs["fast"], ref)]
    worst = max(worst, max(e), max(r))
    print((n, ci, co, h, w, k), "fast vs generic: fwd %.1e dx %.1e dw %.1e db %.1e | vs torch: %.1e %.1e %.1e %.1e" % (*e, *r),
          "<<<<" if max(e + r) > 2e-4 else "")
print("worst", worst)
