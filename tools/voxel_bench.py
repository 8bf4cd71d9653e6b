"""Voxelizer timing on the GPU box: python3 tools/voxel_bench.py [n_events ...]  (per-call ms through the C ABI)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib                                     # noqa: E402
from eemflow_amd.hrem import synthetic_hrem_events               # noqa: E402
from eemflow_amd.voxelizer import EventSequence                  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for n in [int(float(a)) for a in sys.argv[1:]] or [200000, 2000000]:
    for (bins, h, w) in ((5, 720, 1280), (15, 480, 640)):
        ev = synthetic_hrem_events(1, n, h, w)
        seq = EventSequence(None, {"height": h, "width": w}, features=ev, timestamp_multiplier=1e6, convert_to_relative=True)
        feats = torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev)
        grid = torch.empty(bins, h, w, device=dev)
        sp = _lib.current_stream_ptr(dev)
        for norm in (0, 1):
            for _ in range(3):
                _lib.check(L.eemflow_voxelize(feats.data_ptr(), n, bins, h, w, norm, grid.data_ptr(), None, None, sp))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                _lib.check(L.eemflow_voxelize(feats.data_ptr(), n, bins, h, w, norm, grid.data_ptr(), None, None, sp))
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 20
            print("n=%d bins=%d %dx%d normalize=%d: %.3f ms  (%.0f Mev/s)" % (n, bins, h, w, norm, dt * 1e3, n / dt / 1e6), flush=True)
