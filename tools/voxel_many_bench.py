#!/usr/bin/env python3
"""us per SAMPLE (two event sets) of eemflow_voxelize_many for k samples per call, 2 x 10^5 events per set at 1280x720x5 (the
evaluation pipeline's case); EEM_VOX_BAND_FLOATS / EEM_VOX_TWOPASS variants from the environment.  usage: voxel_many_bench.py [nev]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eemflow_amd import _lib
from eemflow_amd.hrem import synthetic_hrem_events
from eemflow_amd.voxelizer import EventSequence

nev = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
H, W = 720, 1280
dev = torch.device("cuda", 0)
L = _lib.lib()
evs = []
for k in range(2):
    seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3 + k, nev, H, W), timestamp_multiplier=1e6, convert_to_relative=True)
    evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
st = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(st.cuda_stream)
res = []
for samples in (1, 2, 5, 10, 16):
    k2 = 2 * samples
    grids = [torch.empty(5, H, W, device=dev) for _ in range(k2)]
    pe = (ctypes.c_void_p * k2)(*[evs[i % 2].data_ptr() for i in range(k2)])
    pn = (ctypes.c_int64 * k2)(*([nev] * k2))
    pg = (ctypes.c_void_p * k2)(*[g.data_ptr() for g in grids])
    for norm in (1, 0):
        for _ in range(5):
            _lib.check(L.eemflow_voxelize_many(k2, pe, pn, 5, H, W, norm, pg, sp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            _lib.check(L.eemflow_voxelize_many(k2, pe, pn, 5, H, W, norm, pg, sp))
        torch.cuda.synchronize()
        res.append((samples, norm, (time.perf_counter() - t0) / n / samples * 1e6))
print("BAND_FLOATS=%s TWOPASS=%s nev=%d: " % (os.environ.get("EEM_VOX_BAND_FLOATS", "-"), os.environ.get("EEM_VOX_TWOPASS", "-"), nev) +
      "  ".join("%d%s: %.1f" % (s, "n" if nm else "r", us) for s, nm, us in res) + "   (us per sample; n = normalised, r = raw)")
