#!/usr/bin/env python3
"""GPU box, stamps build (EEM_BUILD_TAG=voxst EEM_EXTRA_FLAGS=-DEEM_VOX_STAMPS python -m eemflow_amd.build; EEM_LIB_PATH=.../libeemflow_hip_voxst.so):
phase times of vox_band_kernel's first 512 band blocks (job 0) of a 20-grid call at 2e5 events per grid, in us at the 100 MHz s_memtime clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eemflow_amd import _lib
from eemflow_amd.hrem import synthetic_hrem_events
from eemflow_amd.voxelizer import EventSequence

nev, H, W, k2 = 200000, 720, 1280, 20
dev = torch.device("cuda", 0)
L = _lib.lib()
evs = []
for k in range(2):
    seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3 + k, nev, H, W), timestamp_multiplier=1e6, convert_to_relative=True)
    evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
st = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(st.cuda_stream)
grids = [torch.empty(5, H, W, device=dev) for _ in range(k2)]
pe = (ctypes.c_void_p * k2)(*[evs[i % 2].data_ptr() for i in range(k2)])
pn = (ctypes.c_int64 * k2)(*([nev] * k2))
pg = (ctypes.c_void_p * k2)(*[g.data_ptr() for g in grids])
for _ in range(5):
    _lib.check(L.eemflow_voxelize_many(k2, pe, pn, 5, H, W, 0, pg, sp))
torch.cuda.synchronize()
n = 512 * 8
buf = (ctypes.c_ulonglong * n)()
assert L.eemflow_debug_read_vox_stamps(buf, n) == 0
a = np.array(buf[:], dtype=np.int64).reshape(512, 8)
d = np.diff(a[:, :7], axis=1) / 100.0          # s_memtime: 100 MHz
names = ["zero + barrier", "table entries", "first records", "adds + barrier", "read-out (+ stores issued)", "moments"]
print("vox_band_kernel, %d band blocks of job 0: mean (median, p90) us per phase" % len(a))
for i, nm in enumerate(names):
    print("  %-28s %6.2f  (%5.2f, %5.2f)" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
tot = (a[:, 6] - a[:, 0]) / 100.0
print("  block: %.2f (%.2f, %.2f); first start to last end of these blocks %.1f us" % (tot.mean(), np.median(tot), np.percentile(tot, 90), (a[:, 6].max() - a[:, 0].min()) / 100.0))
