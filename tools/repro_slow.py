#!/usr/bin/env python3
"""What slows single-stream E-RAFT after the coalesced pipeline section of bench.py?  usage: repro_slow.py <variant>"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from eemflow_amd import _lib
from eemflow_amd.eraft import ERAFT
from eemflow_amd.eraft_weights import seeded_from_shapes
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair

variant = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda", 0)
H, W = 720, 1280
L = _lib.lib()

def eemflow_ctx():
    sd = seeded_state_dict(0)
    flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in sd.values()]).contiguous()
    c = ctypes.c_void_p()
    _lib.check(L.eemflow_create(0, ctypes.byref(c)))
    _lib.check(L.eemflow_load_weights(c, flat.data_ptr(), flat.numel(), 5, 5))
    _lib.check(L.eemflow_set_image_size(c, H, W, None))
    return c

if variant in ("many", "many_nodestroy", "batch10"):
    c = eemflow_ctx()
    st = torch.cuda.Stream(device=dev)
    n = 10
    e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, n, H, W))
    fl = torch.empty(n, 2, H, W, device=dev)
    with torch.cuda.stream(st):
        sp = ctypes.c_void_p(st.cuda_stream)
        for _ in range(5):
            if variant == "batch10":
                _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), n, H, W, fl.data_ptr(), H, W, sp))
            else:
                arr = ctypes.c_void_p * n
                _lib.check(L.eemflow_forward_many(c, n, arr(*[e1[i].data_ptr() for i in range(n)]), arr(*[e2[i].data_ptr() for i in range(n)]),
                                                  arr(*[fl[i].data_ptr() for i in range(n)]), H, W, H, W, sp))
    torch.cuda.synchronize()
    if variant != "many_nodestroy":
        L.eemflow_destroy(c)
if variant == "vox":
    from eemflow_amd.hrem import synthetic_hrem_events
    from eemflow_amd.voxelizer import EventSequence
    import numpy as np
    seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3, 200000, H, W), timestamp_multiplier=1e6, convert_to_relative=True)
    ev = torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev)
    st = torch.cuda.Stream(device=dev)
    for _ in range(40):
        vv = torch.empty(2, 1, 5, H, W, device=dev)
        _lib.check(L.eemflow_voxelize_pair(ev.data_ptr(), 200000, ev.data_ptr(), 200000, 5, H, W, 1, vv[0].data_ptr(), vv[1].data_ptr(), ctypes.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()

net = ERAFT("", 5).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
net = net.to(dev)
net.change_imagesize((480, 640))
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, 480, 640))
with torch.no_grad():
    for _ in range(3):
        net(e1, e2, iters=12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        net(e1, e2, iters=12)
    torch.cuda.synchronize()
print(variant, "-> E-RAFT b1 %.1f frames/s" % (8 / (time.perf_counter() - t0)))
