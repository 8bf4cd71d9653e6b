"""GPU box: time one convolution shape of the generic conv operator, warm (back to back) and cold (a 1 GB copy between calls evicts
L2 and the infinity cache): tools/conv_time.py n cin cout h w [k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd import ops                                       # noqa: E402

n, ci, co, h, w = (int(v) for v in sys.argv[1:6])
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = "cuda:0"
x = torch.randn(n, ci, h, w, device=dev)
conv = torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
big = torch.empty(256 << 20, device=dev)
big2 = torch.empty_like(big)
with torch.no_grad():
    for _ in range(5):
        ops.conv2d(conv, x, act=ops.ACT_LEAKY)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        ops.conv2d(conv, x, act=ops.ACT_LEAKY)
    ev[1].record()
    torch.cuda.synchronize()
    warm = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    cold = []
    for _ in range(10):
        big2.copy_(big)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.conv2d(conv, x, act=ops.ACT_LEAKY)
        b.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e3)
cold.sort()
print(f"conv {ci}->{co} k{k} {h}x{w} n={n}: warm {warm:.1f} us (repack + conv), cold median {cold[len(cold)//2]:.1f} us, "
      f"{2e-6 * n * h * w * ci * co * k * k / warm:.1f} GFLOP/ms warm")
