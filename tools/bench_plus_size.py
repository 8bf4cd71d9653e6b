"""EEMFlow+ timing at a given batch and size on the GPU box: tools/bench_plus_size.py batch height width"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair
b, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net = EEMFlow_cdc("", 3, 5).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.cuda(); net.change_imagesize((h, w))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, h, w))
with torch.no_grad():
    for _ in range(3): net(e1, e2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): net(e1, e2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"EEMFlow+ {w}x{h} batch={b}: {dt*1e3:.2f} ms/forward, {b/dt:.1f} frames/s")
