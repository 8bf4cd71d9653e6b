#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with EEM_EXTRA_FLAGS=-DEEM_STAMPS): run one forward without the graph and
print the per-phase cycle counts of the LAST wino32 launch (pconv3_3): median / max over blocks of wave 0."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair
L = _lib.lib()
dev = torch.device("cuda", 0)
sd = seeded_state_dict(0)
flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in sd.values()]).contiguous()
c = ctypes.c_void_p(); _lib.check(L.eemflow_create(0, ctypes.byref(c)))
_lib.check(L.eemflow_load_weights(c, flat.data_ptr(), flat.numel(), 5, 5))
H, W = 720, 1280
_lib.check(L.eemflow_set_image_size(c, H, W, None)); _lib.check(L.eemflow_use_graph(c, 0))
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, H, W))
out = torch.empty(1, 2, H, W, device=dev)
for _ in range(3):
    _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), 1, H, W, out.data_ptr(), H, W, None))
torch.cuda.synchronize()
n = 2048 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
L.eemflow_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.eemflow_debug_read_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 8).astype(np.int64)
nb = int((s[:, 0, 0] > 0).sum())
s = s[:nb]
print("blocks", nb)
names = ["start->poff done", "bias+issue(0)", "k loop(+landing)", "barrier+u write", "finish+stores(tile0)", "remaining tiles"]
for w in (0, 7):
    d = np.diff(s[:, w, :7], axis=1)
    print(f"wave {w}: " + "  ".join(f"{nm}: med {int(np.median(d[:, i]))} max {int(d[:, i].max())}" for i, nm in enumerate(names)))
tot = s[:, 0, 6] - s[:, 0, 0]
rt = s[:, 0, 7]
print("total cycles med", int(np.median(tot)), "max", int(tot.max()), " realtime ticks(100MHz) med", int(np.median(rt)),
      " -> clock GHz", round(float(np.median(tot) / np.median(rt) / 10), 3))
t0 = s[:, 0, 0].min()
print("block start spread (cycles)", int(s[:, 0, 0].max() - t0), " last end - first start", int(s[:, :, 6].max() - t0))

# ---- C = 16 kernel (pconv1_2): phases of each block's second tile
fn = getattr(L, "eemflow_debug_read_stamps16", None)
if fn is not None:
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf, n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 8).astype(np.int64)
    nb = int((s[:, 0, 1] > 0).sum())
    s = s[:nb]
    names = ["wait+barrier", "issue next DMA", "k loop", "transform+stores", "pool", ]
    for w in (0, 7):
        d = np.diff(s[:, w, :6], axis=1)
        print(f"C16 wave {w}: " + "  ".join(f"{nm}: med {int(np.median(d[:, i]))} max {int(d[:, i].max())}" for i, nm in enumerate(names)))
    print("C16 kernel cycles med", int(np.median(s[:, 0, 6])), "ticks", int(np.median(s[:, 0, 7])), "GHz",
          round(float(np.median(s[:, 0, 6]) / np.median(s[:, 0, 7]) / 10), 3), "blocks", nb)
