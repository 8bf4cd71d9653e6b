#!/usr/bin/env python3
"""GPU box: bench.py's coalesced pipeline (events -> eemflow_voxelize_many -> eemflow_forward_many -> eemflow_flow_error_many, ten samples per
call, two chains in flight) alone, for a kernel trace: python3 tools/pipeline_trace.py [calls]  -> frames/s; under rocprofv3 --kernel-trace
--stats the per-kernel totals divided by the frame count say what a frame costs in each launch."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eemflow_amd import _lib
from eemflow_amd.weights import seeded_state_dict
from eemflow_amd.hrem import synthetic_hrem_events
from eemflow_amd.voxelizer import EventSequence

H, W, co, nev = 720, 1280, 10, 200000
ncall = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
L = _lib.lib()
flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in seeded_state_dict(0).values()]).to(dev)
evs = []
for k in range(2):
    seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3 + k, nev, H, W), timestamp_multiplier=1e6, convert_to_relative=True)
    evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
gt = torch.from_numpy(np.stack([3 * np.sin(2 * np.pi * xx / W), 3 * np.cos(2 * np.pi * yy / H)])).to(dev)
ctxs, streams = [], []
for _ in range(2):
    cc = ctypes.c_void_p()
    _lib.check(L.eemflow_create(dev.index, ctypes.byref(cc)))
    _lib.check(L.eemflow_load_weights(cc, flat.data_ptr(), flat.numel(), 5, 5))
    _lib.check(L.eemflow_set_image_size(cc, H, W, None))
    _lib.check(L.eemflow_use_graph(cc, 1))
    _lib.check(L.eemflow_set_frames_in_flight(cc, 2))
    _lib.check(L.eemflow_set_deferred_input_norm(cc, 1))
    ctxs.append(cc); streams.append(torch.cuda.Stream(device=dev))
keep = []
mode = os.environ.get("PIPE_MODE", "all")        # all | novox | noerr | fwd


def chain(ci):
    k = ci % 2
    with torch.cuda.stream(streams[k]):
        spk = ctypes.c_void_p(streams[k].cuda_stream)
        vs = [[torch.empty(5 * H * W + 4, device=dev)[:5 * H * W].view(1, 5, H, W) for _ in range(2)] for _ in range(co)]
        fls = [torch.empty(1, 2, H, W, device=dev) for _ in range(co)]
        if mode in ("all", "noerr"):
            k2 = 2 * co
            _lib.check(L.eemflow_voxelize_many(k2, (ctypes.c_void_p * k2)(*[evs[i % 2].data_ptr() for i in range(k2)]), (ctypes.c_int64 * k2)(*([nev] * k2)), 5, H, W, 2,
                                               (ctypes.c_void_p * k2)(*[vs[i // 2][i % 2].data_ptr() for i in range(k2)]), spk))
        arr = ctypes.c_void_p * co
        _lib.check(L.eemflow_forward_many(ctxs[k], co, arr(*[v[0].data_ptr() for v in vs]), arr(*[v[1].data_ptr() for v in vs]), arr(*[f.data_ptr() for f in fls]), H, W, H, W, spk))
        if mode in ("all", "novox"):
            stats = torch.empty(co, 5, device=dev, dtype=torch.float64)
            pa = ctypes.c_void_p * co
            _lib.check(L.eemflow_flow_error_many(co, pa(*[gt.data_ptr()] * co), pa(*[f.data_ptr() for f in fls]), None, H, W, W, stats.data_ptr(), spk))
            keep.append(stats)
        keep.extend(v for pair in vs for v in pair); keep.extend(fls)
        if len(keep) > 8 * co: del keep[:4 * co]


# (novox / fwd: the forward needs grids with a normalisation record: voxelize once into the first buffers it will see is not possible with fresh
# tensors, so those modes run the voxelizer in the warm-up only and time forwards over whatever the fresh tensors hold - timing only)
for ci in range(6): chain(ci)
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for ci in range(ncall): chain(ci)
torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
print("pipeline (%s): %.1f frames/s, %.1f us per frame over %d calls of %d" % (mode, ncall * co / dt, dt / (ncall * co) * 1e6, ncall, co))
