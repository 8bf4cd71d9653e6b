"""Where the evaluation pipeline's time goes (GPU box): python3 tools/pipe_probe.py
events -> eemflow_voxelize x2 -> eemflow_forward -> eemflow_flow_error on three contexts / streams, parts switched on and off, the host's
enqueue time measured beside the wall time, and the same loop driven by one host thread per stream."""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib                                     # noqa: E402
from eemflow_amd.hrem import synthetic_hrem_events               # noqa: E402
from eemflow_amd.voxelizer import EventSequence                  # noqa: E402
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair   # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
H, W, NS, nev = 720, 1280, int(os.environ.get("PIPE_NS", "3")), 200000
flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in seeded_state_dict(0).values()]).contiguous()   # host buffer, as bench.py
e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, H, W))
evs = []
for k in range(2):
    seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3 + k, nev, H, W), timestamp_multiplier=1e6,
                        convert_to_relative=True)
    evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
gt = torch.from_numpy(np.stack([3 * np.sin(2 * np.pi * xx / W), 3 * np.cos(2 * np.pi * yy / H)])).to(dev)
ctxs, streams = [], []
for _ in range(NS):
    cc = ctypes.c_void_p()
    _lib.check(L.eemflow_create(dev.index, ctypes.byref(cc)))
    _lib.check(L.eemflow_load_weights(cc, flat.data_ptr(), flat.numel(), 5, 5))
    _lib.check(L.eemflow_set_image_size(cc, H, W, None))
    _lib.check(L.eemflow_use_graph(cc, 1))
    ctxs.append(cc)
    streams.append(torch.cuda.Stream(device=dev))


def frame(k, vox, fwd, err, keep, norm=1):
    with torch.cuda.stream(streams[k]):
        spk = ctypes.c_void_p(streams[k].cuda_stream)
        v1, v2 = e1, e2
        if vox:
            v1 = torch.empty(1, 5, H, W, device=dev)
            v2 = torch.empty(1, 5, H, W, device=dev)
            _lib.check(L.eemflow_voxelize(evs[0].data_ptr(), nev, 5, H, W, norm, v1.data_ptr(), None, None, spk))
            _lib.check(L.eemflow_voxelize(evs[1].data_ptr(), nev, 5, H, W, norm, v2.data_ptr(), None, None, spk))
        fl = torch.empty(1, 2, H, W, device=dev)
        if fwd:
            _lib.check(L.eemflow_forward(ctxs[k], v1.data_ptr(), v2.data_ptr(), 1, H, W, fl.data_ptr(), H, W, spk))
        if err:
            stats = torch.empty(5, device=dev, dtype=torch.float64)
            _lib.check(L.eemflow_flow_error(gt.data_ptr(), fl.data_ptr(), None, H, W, W, stats.data_ptr(), spk))
            keep.append(stats)
        keep.extend((fl, v1, v2))
        if len(keep) > 48:
            del keep[:16]


def run(name, vox, fwd, err, norm=1, n=300):
    keep = []
    for i in range(30):
        frame(i % NS, vox, fwd, err, keep, norm)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        frame(i % NS, vox, fwd, err, keep, norm)
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    print("%-34s %7.1f us/frame wall   %7.1f us/frame host enqueue   (%.0f frames/s)" % (name, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6, n / (t2 - t0)),
          flush=True)


def run_threads(name, vox, fwd, err, n=300):
    """One host thread per stream (ctypes releases the GIL inside the library calls; each thread has its own voxelizer scratch)."""
    def worker(k, count):
        keep = []
        for _ in range(count):
            frame(k, vox, fwd, err, keep)
    for phase, count in (("warm", 10), ("timed", n // NS)):
        ths = [threading.Thread(target=worker, args=(k, count)) for k in range(NS)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
    m = n // NS * NS
    print("%-34s %7.1f us/frame wall   %7.1f us/frame host enqueue   (%.0f frames/s)" % (name + " [%d threads]" % NS, (t2 - t0) / m * 1e6, (t1 - t0) / m * 1e6,
                                                                                      m / (t2 - t0)), flush=True)


vstream = torch.cuda.Stream(device=dev)


def frame_decoupled(i, keep):
    """The voxelizer on its own stream (the data loader's place, ahead of the model), the model streams wait for the volumes' event."""
    k = i % NS
    with torch.cuda.stream(vstream):
        spv = ctypes.c_void_p(vstream.cuda_stream)
        v1 = torch.empty(1, 5, H, W, device=dev)
        v2 = torch.empty(1, 5, H, W, device=dev)
        _lib.check(L.eemflow_voxelize(evs[0].data_ptr(), nev, 5, H, W, 1, v1.data_ptr(), None, None, spv))
        _lib.check(L.eemflow_voxelize(evs[1].data_ptr(), nev, 5, H, W, 1, v2.data_ptr(), None, None, spv))
        ready = torch.cuda.Event()
        ready.record(vstream)
    with torch.cuda.stream(streams[k]):
        spk = ctypes.c_void_p(streams[k].cuda_stream)
        streams[k].wait_event(ready)
        v1.record_stream(streams[k])
        v2.record_stream(streams[k])
        fl = torch.empty(1, 2, H, W, device=dev)
        _lib.check(L.eemflow_forward(ctxs[k], v1.data_ptr(), v2.data_ptr(), 1, H, W, fl.data_ptr(), H, W, spk))
        stats = torch.empty(5, device=dev, dtype=torch.float64)
        _lib.check(L.eemflow_flow_error(gt.data_ptr(), fl.data_ptr(), None, H, W, W, stats.data_ptr(), spk))
        keep.extend((fl, v1, v2, stats))
        if len(keep) > 64:
            del keep[:16]


def run_decoupled(n=300):
    keep = []
    for i in range(30):
        frame_decoupled(i, keep)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        frame_decoupled(i, keep)
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    print("%-34s %7.1f us/frame wall   %7.1f us/frame host enqueue   (%.0f frames/s)" % ("full pipeline, voxelizer stream", (t2 - t0) / n * 1e6,
                                                                                      (t1 - t0) / n * 1e6, n / (t2 - t0)), flush=True)


run("forward (fresh output)", 0, 1, 0)
run("forward + flow_error", 0, 1, 1)
run("voxelize x2 only", 1, 0, 0)
run("voxelize x2 only, no normalise", 1, 0, 0, norm=0)
run("voxelize x2 + forward", 1, 1, 0)
run("full pipeline", 1, 1, 1)
run_threads("forward (fresh output)", 0, 1, 0)
run_threads("full pipeline", 1, 1, 1)
run_decoupled()
run("full pipeline", 1, 1, 1)
run_decoupled()
for cc in ctxs:
    L.eemflow_destroy(cc)
