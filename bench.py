#!/usr/bin/env python3
"""EEMFlow hot-path benchmark on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) or on its
  own - then bench.py starts that launcher itself as a child process before it touches the GPU (self_launch below).

A step is one EEMFlow inference forward (libeemflow_hip.so, HIP-graph replay) over one batch of
synthetic event-voxel pairs already resident in HBM: BASELINE.json configs[1], 1280x720, batch 1.
Ranks are independent replicas (frames shard over GPUs, no data-path collective): weak scaling.
Rank 0 prints ONE JSON line with the whole-job frames/s, the roofline of the dominant kernel
(per-kernel HIP-event timing via eemflow_time_kernels) and the CPU baseline (the oracle - the
reference's PyTorch-CPU path restated - timed on this host's cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them the null stream's): the fourth
# frame in flight then shares a queue with another and serialises behind it (5 700 frames/s with four streams against 6 350 with three;
# with its own queue 6 670).  Read at the runtime's first call, so it is set before torch touches the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np   # noqa: E402
import torch         # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec peak
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak (same guide)
# an fp32 product computed as six bf16-piece MFMAs (conv_bx3.hip): the roof such a kernel runs against, in fp32-equivalent FLOPs
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 6.0
ROTATE_PAIRS = 8                 # distinct input pairs walked by the timed loop: 8 x 36.9 MB > the 256 MiB Infinity Cache at 1280x720
N_CUS = 256


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--batch", type=int, default=1)
    p.add_argument("--height", type=int, default=720)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    p.add_argument("--kernel-reps", type=int, default=20)
    p.add_argument("--preheat", type=int, default=200, help="untimed forwards before the warm-up steps (GPU clocks, graph-launch paths)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-other-rows", action="store_true", help="skip the E-RAFT / training-step side timings")
    p.add_argument("--no-side-rows", action="store_true",
                   help="only the timed loop and the per-kernel table of its launch configuration (the rocprofv3 runs: their kernel "
                        "averages then belong to one configuration)")
    p.add_argument("--mode", choices=("infer", "train"), default="infer",
                   help="infer (default): the headline metric; train: BASELINE configs[3] - EEMFlow training step, 1280x720, batch 8 per GPU, "
                        "data parallel with one RCCL all-reduce of the flat gradient per step (its time is reported as allreduce_us)")
    p.add_argument("--workload", choices=("eemflow", "eraft"), default="eemflow",
                   help="eemflow (default): BASELINE configs[1] (infer) / configs[3] (train); eraft: BASELINE configs[4] - E-RAFT 640x480, 12 "
                        "refinement iterations, batch 4 per GPU, inference (--mode infer) or the training step with one all-reduce of the "
                        "flat parameter gradient per step (--mode train)")
    p.add_argument("--iters", type=int, default=12, help="--workload eraft: refinement iterations")
    p.add_argument("--frames-in-flight", type=int, default=0,
                   help="eemflow_set_frames_in_flight hint for the contexts (default: --streams); the profiles use --streams 1 "
                        "--frames-in-flight 4 to trace the timed loop's launch configuration one kernel at a time")
    p.add_argument("--rotate-pairs", type=int, default=ROTATE_PAIRS,
                   help="distinct input pairs the timed loop walks (1: the same pair every step, an Infinity-Cache-resident input)")
    p.add_argument("--streams", type=int, default=0,
                   help="chains of launches in flight per GPU: independent contexts on separate HIP streams, calls alternate "
                        "(default: 2 with --coalesce > 1, else 4)")
    p.add_argument("--coalesce", type=int, default=10,
                   help="independent batch-1 frames handed to the library per call (eemflow_forward_many: n frames in n unrelated buffers ride "
                        "one batch-n chain of launches); 1 = one eemflow_forward per frame.  A step stays ONE frame at batch 1")
    p.add_argument("--rows-child", action="store_true",
                   help="internal: measure the other rows (E-RAFT, EEMFlow+, voxelizer, training steps) in THIS fresh process and print them as one JSON line")
    p.add_argument("--long-steps", type=int, default=400,
                   help="a second, longer run of the same timed loop after the K steps (reported as value_long; 0 disables)")
    return p.parse_args()


def usable_cores():
    """Logical CPUs this process may use: affinity mask, capped by a cgroup-v2 CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sd_np, e1, e2, budget_s):
    """The oracle (PyTorch-CPU restatement of the reference path) on the host cores, bounded sample.
    The thread count is the fastest of a short sweep (all usable cores is not always best on a
    2-socket SMT host); `cores` reports the threads actually used."""
    from oracle import eemflow_oracle as O
    sd = O.to_torch_sd(sd_np)
    ncpu = usable_cores()
    cands = sorted({c for c in (ncpu, ncpu // 2, 64, 32, 16, 8) if 1 <= c <= ncpu}, reverse=True)
    sweep = {}
    with torch.no_grad():
        ref, _ = O.eemflow_forward(sd, e1, e2)
        def med3(c):
            torch.set_num_threads(c)
            O.eemflow_forward(sd, e1, e2)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                O.eemflow_forward(sd, e1, e2)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts))
        for c in cands:
            sweep[c] = med3(c)
            if sweep[c] > 5.0 and len(sweep) >= 2:
                break
        single = sweep[1] if 1 in sweep else med3(1)     # the single-thread figure beside the all-cores one
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        O.eemflow_forward(sd, e1, e2)
        times = []
        t_start = time.perf_counter()
        while (time.perf_counter() - t_start < budget_s and len(times) < 2000) or len(times) < 3:
            t0 = time.perf_counter()
            O.eemflow_forward(sd, e1, e2)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return ref, {"value": e1.shape[0] / med, "unit": "frames/s", "cores": best, "kind": "port",
                 "sample": f"{len(times)} forwards of the same {e1.shape[0]}x5x{e1.shape[2]}x{e1.shape[3]} pair, "
                           f"median {med * 1e3:.1f} ms, torch {torch.__version__} CPU fp32, {best} threads "
                           f"(fastest of a sweep of medians of three: {{{', '.join(f'{k}: {v * 1e3:.0f} ms' for k, v in sweep.items())}}}; "
                           f"host has {os.cpu_count()} logical CPUs, {ncpu} usable); one thread: {single * 1e3:.0f} ms",
                 "ms_per_frame": med * 1e3 / e1.shape[0],
                 "single_thread_value": e1.shape[0] / single, "single_thread_ms_per_frame": single * 1e3 / e1.shape[0]}


def other_rows(dev):
    """Short timings of the other built rows of the scope table (not the headline metric): E-RAFT inference at
    BASELINE configs[4]'s shape and the EEMFlow training step at configs[2]'s shape.  Never fatal."""
    from eemflow_amd import _lib as _lib_mod
    out = {}
    try:
        from eemflow_amd.eraft import ERAFT
        from eemflow_amd.eraft_weights import seeded_from_shapes
        from eemflow_amd.weights import synthetic_voxel_pair
        net = ERAFT("", 5).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in
                             seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
        net = net.to(dev)
        net.change_imagesize((480, 640))
        e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, 480, 640))
        with torch.no_grad():
            for _ in range(3):                                   # (workspace growth, weight packing, clocks: three untimed forwards)
                net(e1, e2, iters=12)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(8):
                net(e1, e2, iters=12)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
        out["eraft_640x480_12it_b1_frames_per_s"] = round(8 / (time.perf_counter() - t0), 2)
        out["eraft_640x480_12it_b1_host_enqueue_ms"] = round(t_enq / 8 * 1e3, 2)
        e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 4, 480, 640))       # configs[4]: batch 4 per GPU
        with torch.no_grad():
            for _ in range(2):
                net(e1, e2, iters=12)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                net(e1, e2, iters=12)
            torch.cuda.synchronize(dev)
        out["eraft_640x480_12it_b4_frames_per_s"] = round(20 / (time.perf_counter() - t0), 2)
        # the evaluation loop's use (test_mvsec.py:1455 reads flow_list[-1]): ERAFT.final_only - same last prediction, the mask head and
        # the convex upsampling of the eleven iterations before it not launched.  Rows of their own; the two above form all twelve.
        net.final_only = True
        for key, batch, reps in (("eraft_640x480_12it_b1_final_only_frames_per_s", 1, 8), ("eraft_640x480_12it_b4_final_only_frames_per_s", 4, 5)):
            e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, batch, 480, 640))
            with torch.no_grad():
                for _ in range(2):
                    net(e1, e2, iters=12)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    net(e1, e2, iters=12)
                torch.cuda.synchronize(dev)
            out[key] = round(reps * batch / (time.perf_counter() - t0), 2)
        net.final_only = False
        # the evaluation loop's independent batch-1 samples, four per call, each in its own tensors (ERAFT.forward_many)
        frames4 = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(20 + i, 1, 480, 640)) for i in range(4)]
        with torch.no_grad():
            for _ in range(2):
                net.forward_many(frames4, iters=12)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                net.forward_many(frames4, iters=12)
            torch.cuda.synchronize(dev)
        out["eraft_640x480_12it_b1_coalesced4_frames_per_s"] = round(20 / (time.perf_counter() - t0), 2)
        del frames4
        # several frames in flight (one module / context per HIP stream): at batch 1 the 60x80 update block launches ~300 blocks for
        # 256 CUs - a second and third frame fill the chip
        nets = [net]
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        for _ in range(3):
            m = ERAFT("", 5).eval()
            m.load_state_dict(sd)
            m = m.to(dev)
            m.change_imagesize((480, 640))
            nets.append(m)
        for m in nets:
            m.frames_in_flight = 4
        streams = [torch.cuda.Stream(device=dev) for _ in nets]
        for key, batch, ns in (("eraft_640x480_12it_b1_4_in_flight_frames_per_s", 1, 4), ("eraft_640x480_12it_b4_3_in_flight_frames_per_s", 4, 3)):
            e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, batch, 480, 640))
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for phase, n in (("warm", ns), ("timed", 4 * ns)):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for i in range(n):
                        with torch.cuda.stream(streams[i % ns]):
                            nets[i % ns](e1, e2, iters=12)
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
            out[key] = round(4 * ns * batch / dt, 2)
        del nets, net
    except Exception as e:                                   # noqa: BLE001
        out["eraft_error"] = repr(e)[:200]
    try:
        from eemflow_amd.eemflow_plus import EEMFlow_cdc
        from eemflow_amd.plus_weights import seeded_from_shapes as plus_seeded
        from eemflow_amd.weights import synthetic_voxel_pair
        net = EEMFlow_cdc("", 3, 5).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in
                             plus_seeded({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
        net = net.to(dev)
        net.change_imagesize((720, 1280))
        e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 1, 720, 1280))
        with torch.no_grad():
            for _ in range(3):
                net(e1, e2)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                net(e1, e2)
            torch.cuda.synchronize(dev)
        out["eemflow_plus_1280x720_b1_frames_per_s"] = round(20 / (time.perf_counter() - t0), 2)
        # the evaluation loop's independent batch-1 samples, four per call, each in its own tensors (EEMFlow_cdc.forward_many)
        frames4 = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(10 + i, 1, 720, 1280)) for i in range(4)]
        with torch.no_grad():
            for _ in range(2):
                net.forward_many(frames4)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(6):
                net.forward_many(frames4)
            torch.cuda.synchronize(dev)
        out["eemflow_plus_1280x720_b1_coalesced4_frames_per_s"] = round(24 / (time.perf_counter() - t0), 2)
        del frames4
        nets = [net]
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        for _ in range(3):                                   # four frames in flight: ~160 short launches per frame leave CUs idle
            m = EEMFlow_cdc("", 3, 5).eval()
            m.load_state_dict(sd)
            m = m.to(dev)
            m.change_imagesize((720, 1280))
            nets.append(m)
        for m in nets:
            m.frames_in_flight = 4
        streams = [torch.cuda.Stream(device=dev) for _ in nets]
        with torch.no_grad():
            for phase, n in (("warm", 4), ("timed", 24)):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(n):
                    with torch.cuda.stream(streams[i % 4]):
                        nets[i % 4](e1, e2)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
        out["eemflow_plus_1280x720_b1_4_in_flight_frames_per_s"] = round(24 / dt, 2)
        del nets, net
    except Exception as e:                                   # noqa: BLE001
        out["eemflow_plus_error"] = repr(e)[:200]
    try:
        from eemflow_amd.hrem import synthetic_hrem_events
        from eemflow_amd.voxelizer import EventSequence, EventSequenceToVoxelGrid_Pytorch
        ev = synthetic_hrem_events(1, 2000000, 720, 1280)
        seq = EventSequence(None, {"height": 720, "width": 1280}, features=ev, timestamp_multiplier=1e6, convert_to_relative=True)
        feats = torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev)
        grid = torch.empty(5, 720, 1280, device=dev)
        L = _lib_mod.lib()
        sp = _lib_mod.current_stream_ptr(dev)
        for _ in range(3):
            _lib_mod.check(L.eemflow_voxelize(feats.data_ptr(), feats.shape[0], 5, 720, 1280, 1, grid.data_ptr(), None, None, sp))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            _lib_mod.check(L.eemflow_voxelize(feats.data_ptr(), feats.shape[0], 5, 720, 1280, 1, grid.data_ptr(), None, None, sp))
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 20
        out["voxelize_2M_events_1280x720_ms"] = round(dt * 1e3, 3)
        out["voxelize_Mevents_per_s"] = round(2.0 / dt, 1)
        grid2 = torch.empty(5, 720, 1280, device=dev)
        for nn, key in ((2000000, "voxelize_pair_2x2M_events_ms"), (200000, "voxelize_pair_2x200k_events_ms")):
            for _ in range(3):
                _lib_mod.check(L.eemflow_voxelize_pair(feats.data_ptr(), nn, feats.data_ptr(), nn, 5, 720, 1280, 1, grid.data_ptr(), grid2.data_ptr(), sp))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                _lib_mod.check(L.eemflow_voxelize_pair(feats.data_ptr(), nn, feats.data_ptr(), nn, 5, 720, 1280, 1, grid.data_ptr(), grid2.data_ptr(), sp))
            torch.cuda.synchronize(dev)
            out[key] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    except Exception as e:                                   # noqa: BLE001
        out["voxelize_error"] = repr(e)[:200]
    try:
        from eemflow_amd import EEMFlow
        from eemflow_amd.train import EEMFlowTrainer
        from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
        b, h, w = 32, 260, 346
        net = EEMFlow("", 5, 5)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
        net = net.to(dev).train()
        net.change_imagesize((h, w))
        tr = EEMFlowTrainer(net, lr=1e-4, num_steps=1000)
        e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, b, h, w))
        gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, b, h, w))
        for _ in range(3):
            tr.step(e1, e2, gt, va)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(e1, e2, gt, va)
        torch.cuda.synchronize(dev)
        out["train_step_346x260_b32_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
    except Exception as e:                                   # noqa: BLE001
        out["train_error"] = repr(e)[:200]
    try:                                                     # E-RAFT training step at configs[4]'s shape (operator-level autograd route)
        from eemflow_amd.eraft import ERAFT
        from eemflow_amd.eraft_weights import seeded_from_shapes
        from eemflow_amd.train import sequence_loss
        from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
        net = ERAFT("", 5)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in
                             seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
        net = net.to(dev).train()
        net.change_imagesize((480, 640))
        opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8)
        e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1, 4, 480, 640))
        gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(2, 4, 480, 640))

        def estep():
            opt.zero_grad()
            loss, _ = sequence_loss(net(e1, e2, iters=12)[1], gt, va, 0.8)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
            opt.step()
        estep()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(2):
            estep()
        torch.cuda.synchronize(dev)
        out["eraft_train_step_640x480_12it_b4_ms"] = round((time.perf_counter() - t0) / 2 * 1e3, 1)
        del net, opt
    except Exception as e:                                   # noqa: BLE001
        out["eraft_train_error"] = repr(e)[:200]
    return out


def latency_and_pipeline_rows(L, _lib, dev, flat, B, H, W, e1, e2, use_graph, coalesce=1):
    """Two rows beside the headline (not the metric): single-stream latency of one frame, and the evaluation loop of
    test_mvsec.py:580-597 as a device-resident pipeline with FRESH tensors every frame: events -> eemflow_voxelize x2 -> forward ->
    eemflow_flow_error on four contexts / streams (the graph cache is keyed on shapes, so new buffers replay the same graph)."""
    out = {}
    c = ctypes.c_void_p()
    _lib.check(L.eemflow_create(dev.index, ctypes.byref(c)))
    _lib.check(L.eemflow_load_weights(c, flat.data_ptr(), flat.numel(), 5, 5))
    _lib.check(L.eemflow_set_image_size(c, H, W, None))
    _lib.check(L.eemflow_use_graph(c, 1 if use_graph else 0))
    st = torch.cuda.Stream(device=dev)
    sp = ctypes.c_void_p(st.cuda_stream)
    flow = torch.empty(B, 2, H, W, device=dev)
    for _ in range(10):
        _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), B, H, W, flow.data_ptr(), H, W, sp))
    torch.cuda.synchronize(dev)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), B, H, W, flow.data_ptr(), H, W, sp))
    torch.cuda.synchronize(dev)
    out["latency_ms_b1"] = round((time.perf_counter() - t0) / n * 1e3, 4)          # frames back to back on ONE stream
    t0 = time.perf_counter()
    for _ in range(50):
        _lib.check(L.eemflow_forward(c, e1.data_ptr(), e2.data_ptr(), B, H, W, flow.data_ptr(), H, W, sp))
        st.synchronize()
    out["latency_ms_b1_synchronised"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)   # host waits for every frame
    L.eemflow_destroy(c)
    if B != 1:
        return out
    try:
        from eemflow_amd.hrem import synthetic_hrem_events
        from eemflow_amd.voxelizer import EventSequence
        NS, nev = 4, 200000
        evs = []
        for k in range(2):
            seq = EventSequence(None, {"height": H, "width": W}, features=synthetic_hrem_events(3 + k, nev, H, W),
                                timestamp_multiplier=1e6, convert_to_relative=True)
            evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        gt = torch.from_numpy(np.stack([3 * np.sin(2 * np.pi * xx / W), 3 * np.cos(2 * np.pi * yy / H)])).to(dev)
        ctxs, streams = [], []
        for _ in range(NS):
            cc = ctypes.c_void_p()
            _lib.check(L.eemflow_create(dev.index, ctypes.byref(cc)))
            _lib.check(L.eemflow_load_weights(cc, flat.data_ptr(), flat.numel(), 5, 5))
            _lib.check(L.eemflow_set_image_size(cc, H, W, None))
            _lib.check(L.eemflow_use_graph(cc, 1 if use_graph else 0))
            _lib.check(L.eemflow_set_frames_in_flight(cc, NS))
            ctxs.append(cc)
            streams.append(torch.cuda.Stream(device=dev))
        keep = []

        def frame(i, voxelize):
            k = i % NS
            with torch.cuda.stream(streams[k]):
                spk = ctypes.c_void_p(streams[k].cuda_stream)
                if voxelize:
                    vv = torch.empty(2, 1, 5, H, W, device=dev)             # both volumes of the sample by one launch sequence
                    v1, v2 = vv[0], vv[1]
                    _lib.check(L.eemflow_voxelize_pair(evs[0].data_ptr(), nev, evs[1].data_ptr(), nev, 5, H, W, 1, v1.data_ptr(), v2.data_ptr(), spk))
                else:
                    v1, v2 = e1, e2
                fl = torch.empty(1, 2, H, W, device=dev)                     # fresh output tensor every frame
                _lib.check(L.eemflow_forward(ctxs[k], v1.data_ptr(), v2.data_ptr(), 1, H, W, fl.data_ptr(), H, W, spk))
                if voxelize:
                    stats = torch.empty(5, device=dev, dtype=torch.float64)
                    _lib.check(L.eemflow_flow_error(gt.data_ptr(), fl.data_ptr(), None, H, W, W, stats.data_ptr(), spk))
                    keep.append(stats)
                keep.append(fl)
                if len(keep) > 24:                                          # drop old tensors: the allocator hands out other blocks
                    del keep[:8]
        for mode, key in ((False, "fresh_buffers_frames_per_s"), (True, "pipeline_one_frame_per_call_frames_per_s")):
            for i in range(30):
                frame(i, mode)
            torch.cuda.synchronize(dev)
            n = 300
            t0 = time.perf_counter()
            for i in range(n):
                frame(i, mode)
            torch.cuda.synchronize(dev)
            out[key] = round(n / (time.perf_counter() - t0), 1)

        # the same pipeline with the timed loop's coalescing: `co` samples are voxelized (one pair call each), handed to ONE
        # eemflow_forward_many call - every sample in its own fresh tensors - and scored (flow_error each), two such chains in flight
        def chain(ci, co):
            k = ci % 2
            with torch.cuda.stream(streams[k]):
                spk = ctypes.c_void_p(streams[k].cuda_stream)
                vs, fls = [], []
                for _ in range(co):                              # two raw grids per sample, each with room for its normalisation record
                    vs.append([torch.empty(5 * H * W + 4, device=dev)[:5 * H * W].view(1, 5, H, W) for _ in range(2)])
                    fls.append(torch.empty(1, 2, H, W, device=dev))
                # both volumes of the co samples by ONE voxelizer launch sequence (up to 32 event sets per call)
                for s0 in range(0, co, 16):
                    part = vs[s0:s0 + 16]
                    k2 = 2 * len(part)
                    _lib.check(L.eemflow_voxelize_many(k2, (ctypes.c_void_p * k2)(*[evs[i % 2].data_ptr() for i in range(k2)]),
                                                       (ctypes.c_int64 * k2)(*([nev] * k2)), 5, H, W, 2,
                                                       (ctypes.c_void_p * k2)(*[part[i // 2][i % 2].data_ptr() for i in range(k2)]), spk))
                arr = ctypes.c_void_p * co
                _lib.check(L.eemflow_forward_many(ctxs[k], co, arr(*[v[0].data_ptr() for v in vs]), arr(*[v[1].data_ptr() for v in vs]),
                                                  arr(*[f.data_ptr() for f in fls]), H, W, H, W, spk))
                for s0 in range(0, co, 16):                      # the call's samples scored by ONE launch (up to 16 per call)
                    part = fls[s0:s0 + 16]
                    stats = torch.empty(len(part), 5, device=dev, dtype=torch.float64)
                    pa = ctypes.c_void_p * len(part)
                    _lib.check(L.eemflow_flow_error_many(len(part), pa(*[gt.data_ptr()] * len(part)), pa(*[f.data_ptr() for f in part]), None,
                                                         H, W, W, stats.data_ptr(), spk))
                    keep.append(stats)
                keep.extend(v for pair in vs for v in pair)
                keep.extend(fls)
                if len(keep) > 8 * co:
                    del keep[:4 * co]
        for cc in ctxs[:2]:
            _lib.check(L.eemflow_set_frames_in_flight(cc, 2))
            _lib.check(L.eemflow_set_deferred_input_norm(cc, 1))     # normalisation applied by pconv1_1 as it reads the raw grids
        co = max(1, coalesce)
        del keep[:]
        _skip = os.environ.get("EEM_BENCH_SKIP", "")
        if "chain" in _skip:
            co = 1
        for ci in range(4):
            chain(ci, co)
        torch.cuda.synchronize(dev)
        ncall = max(4, 300 // co)
        t0 = time.perf_counter()
        for ci in range(ncall):
            chain(ci, co)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        out["pipeline_frames_per_s"] = round(ncall * co / (time.perf_counter() - t0), 1)
        out["pipeline_host_enqueue_us_per_frame"] = round(t_enq / (ncall * co) * 1e6, 1)
        out["pipeline_coalesced_frames"] = co
        out["pipeline_normalisation"] = "deferred to pconv1_1 (voxelizer normalize=2)"
        for cc in ctxs[:2]:
            _lib.check(L.eemflow_set_deferred_input_norm(cc, 0))
        del keep[:]
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()                             # the side rows below allocate through hipMalloc: hand the pipeline's blocks back first
        # the same frames as batches of four per forward, two forwards in flight (B in {2, 8, 32} for throughput): more tiles
        # per persistent block - NOT the headline configuration (batch 1 per forward), reported beside it
        from eemflow_amd.weights import synthetic_voxel_pair as _svp
        if "batch4" in _skip:
            raise RuntimeError("skipped")
        b1, b2 = (torch.from_numpy(a).to(dev) for a in _svp(0, 4, H, W))
        fb = [torch.empty(4, 2, H, W, device=dev) for _ in range(2)]
        for cc in ctxs[:2]:
            _lib.check(L.eemflow_set_frames_in_flight(cc, 2))
        for phase, n in (("warm", 6), ("timed", 60)):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n):
                k = i % 2
                _lib.check(L.eemflow_forward(ctxs[k], b1.data_ptr(), b2.data_ptr(), 4, H, W, fb[k].data_ptr(), H, W,
                                             ctypes.c_void_p(streams[k].cuda_stream)))
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        out["batch4_two_in_flight_frames_per_s"] = round(60 * 4 / dt, 1)
        gs = (ctypes.c_longlong * 3)()
        _lib.check(L.eemflow_graph_stats(ctxs[0], ctypes.byref(gs)))
        out["pipeline_graph_captures_replays_io_updates_ctx0"] = list(gs)
        out["pipeline_events_per_volume"] = nev
        for cc in ctxs:
            L.eemflow_destroy(cc)
    except Exception as e:                                   # noqa: BLE001
        out["pipeline_error"] = repr(e)[:200]
    return out


def baseline_metric():
    """BASELINE.json's metric string (the file travels with the repo); a literal copy if it is missing."""
    try:
        return json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"]
    except Exception:
        return "frames/sec + EPE, EEMFlow 1280\u00d7720 dt1, 1/2/4/8 MI355X"


def csrc_sha():
    """sha256 over the kernel sources (the same digest tools/pmc_traffic.py stamps its measurement with)."""
    import hashlib
    root = os.path.join(REPO, "eemflow_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """profiles/pmc_traffic.json - HBM bytes per launch from rocprofv3 PMC passes (tools/profile_gpu.sh).  Quoted only while the kernel
    sources are the ones it was measured on: after any change to csrc/ the file is stale and `traffic` is null until the passes are
    re-run."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(path):
        try:
            t = json.load(open(path))
        except Exception:
            return None
        if t.get("_meta", {}).get("csrc_sha") != csrc_sha():
            return {"_stale": True, "_meta": t.get("_meta", {})}
        return t
    return None


def main_train(args):
    """BASELINE configs[3] (configs[2] with --height 260 --width 346 --batch 32): one optimisation step per `step` - forward with
    activations kept, sequence loss, backward into the flat gradient (eemflow_forward_backward), ONE all-reduce of that 2.86 MB buffer
    over RCCL + division by the world size (eemflow_amd.parallel.average_gradients), clip + AdamW + weight re-pack
    (eemflow_optimizer_step).  value = samples of ALL ranks per second; allreduce_us = HIP-event time of the collective per step
    on rank 0's stream (0 at N = 1)."""
    from eemflow_amd import EEMFlow, _lib, parallel
    from eemflow_amd.train import OneCycleLinear
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    rank, local_rank, world = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    dev = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    if world > 1:
        parallel.pin_host_threads_to_gpu_numa(dev.index)         # one rank per GPU: its host threads next to that GPU
    B = args.batch if args.batch > 1 else 8
    H, W = args.height, args.width
    L = _lib.lib()
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})      # same start on every rank
    net = net.to(dev).train()
    net.change_imagesize((H, W))
    ctx = net._context(dev)
    e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1 + rank, B, H, W))    # each rank its own shard
    gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(100 + rank, B, H, W))
    n = sum(p.numel() for p in net.parameters())
    grad = torch.empty(n, device=dev)
    flow = torch.empty(B, 2, H, W, device=dev)
    sched = OneCycleLinear(1e-4, args.steps + args.warmup + 100)
    sp = _lib.current_stream_ptr(dev)
    evs = []

    def step(i, timed):
        stats = (ctypes.c_double * 5)()
        _lib.check(L.eemflow_forward_backward(ctx, e1.data_ptr(), e2.data_ptr(), gt.data_ptr(), va.data_ptr(), B, H, W, H, W, 1.0,
                                              flow.data_ptr(), grad.data_ptr(), ctypes.byref(stats), sp))
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        parallel.average_gradients(grad)
        if timed:
            b.record()
            evs.append((a, b))
        _lib.check(L.eemflow_optimizer_step(ctx, grad.data_ptr(), sched.lr(i), 5e-5, 1e-8, 1.0, sp))
        return stats[0]

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize(dev)
    parallel.barrier(dev)
    t0 = time.perf_counter()
    loss = 0.0
    for i in range(args.steps):
        loss = step(args.warmup + i, True)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    parallel.barrier(dev)
    value, slowest = parallel.aggregate_throughput(args.steps * B, elapsed, dev)
    ar_us = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3 if evs else 0.0
    if rank == 0:
        flops = 3 * 14.583e9 * (H * W) / (720 * 1280)                   # ~3x the forward's direct-convolution count per sample
        line = {"metric": f"samples/sec, EEMFlow training step {W}x{H}, batch {B}/GPU, data parallel (BASELINE configs[3] shape)",
                "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(slowest * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"EEMFlow training step, {W}x{H}, batch={B} per GPU: forward + sequence loss + backward + "
                                       "gradient all-reduce + clip + AdamW + re-pack", "height": H, "width": W, "batch_per_gpu": B,
                           "parallelism": f"dp{world}: batch sharded over ranks, one RCCL all-reduce of {n * 4} gradient bytes per step"},
                "allreduce_us": round(ar_us, 1), "allreduce_ms": round(ar_us / 1e3, 4), "allreduce_bytes": n * 4, "final_loss": loss,
                "tflops_per_gpu_at_3x_forward": round(flops * B / (slowest / args.steps) / 1e12, 2)}
        print(json.dumps(line), flush=True)
    parallel.barrier(dev)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


ERAFT_GFLOP_PER_FRAME_640x480_12IT = 499.2      # direct-convolution + correlation FLOPs of one 640x480 frame, 12 iterations


def main_eraft(args):
    """BASELINE configs[4]: E-RAFT (model/eraft.py:97-159) at 640x480, 12 refinement iterations, batch 4 per GPU.
    --mode infer: a step = one forward of this rank's batch (frames shard over ranks, no data-path collective; weak scaling).
    --mode train: a step = train_mvsec.py:241-258's statement sequence on the operator-level autograd route - forward, sequence loss,
    backward, ONE all-reduce of the flat parameter gradient (parallel.average_gradients: RCCL over xGMI on a multi-GPU node), clip,
    AdamW - the exchange harness.TrainRaftEvents._train_iters_autograd makes; allreduce_us = its HIP-event time on rank 0."""
    from eemflow_amd import parallel
    from eemflow_amd.eraft import ERAFT
    from eemflow_amd.eraft_weights import seeded_from_shapes
    from eemflow_amd.train import sequence_loss
    from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
    rank, local_rank, world = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    dev = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    if world > 1:
        parallel.pin_host_threads_to_gpu_numa(dev.index)         # one rank per GPU: its host threads next to that GPU
    B = args.batch if args.batch > 1 else 4
    H, W = (480, 640) if (args.height, args.width) == (720, 1280) else (args.height, args.width)
    iters = args.iters
    train = args.mode == "train"
    net = ERAFT("", 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in
                         seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})   # same start on every rank
    net = net.to(dev)
    net = net.train() if train else net.eval()
    net.change_imagesize((H, W))
    e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(1 + rank, B, H, W))          # each rank its own frames
    evs = []
    if train:
        gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(100 + rank, B, H, W))
        opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=5e-5, eps=1e-8)
        params = [p for p in net.parameters() if p.requires_grad]
        nparam = sum(p.numel() for p in params)
        last = {"loss": 0.0}

        def step(timed):
            opt.zero_grad()
            loss, _ = sequence_loss(net(e1, e2, iters=iters)[1], gt, va, 0.8, metrics=False)
            loss.backward()
            have = [p for p in params if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in have])
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            parallel.average_gradients(flat)
            if timed:
                b.record()
                evs.append((a, b))
            if parallel.exchange_active():
                off = 0
                for p in have:
                    p.grad.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
            opt.step()
            last["loss"] = loss
    else:
        nparam = sum(p.numel() for p in net.parameters())

        def step(timed):
            with torch.no_grad():
                net(e1, e2, iters=iters)

    for _ in range(max(args.warmup, 1)):
        step(False)
    torch.cuda.synchronize(dev)
    parallel.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    parallel.barrier(dev)
    value, slowest = parallel.aggregate_throughput(args.steps * B, elapsed, dev)
    ar_us = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3 if evs else 0.0
    if rank == 0:
        ms_per_step = slowest * 1e3 / args.steps
        gflop = ERAFT_GFLOP_PER_FRAME_640x480_12IT * (H * W) / (480 * 640) * iters / 12 * (3 if train else 1)     # ~3x forward per training sample
        what = "training step" if train else "inference"
        line = {"metric": f"{'samples' if train else 'frames'}/sec, E-RAFT {what} {W}x{H}, {iters} refinement iterations, batch {B}/GPU (BASELINE configs[4])",
                "value": round(value, 2), "unit": "samples/s" if train else "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": max(args.warmup, 1), "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"E-RAFT {what}, {W}x{H}, {iters} refinement iterations, batch={B} per GPU (BASELINE configs[4]); "
                                       "synthetic voxel pairs resident in HBM, seeded weights",
                           "height": H, "width": W, "batch_per_gpu": B, "iters": iters,
                           "parallelism": (f"dp{world}: batch sharded over ranks, one all-reduce of {nparam * 4} parameter-gradient bytes per step"
                                           if train else f"replicas x{world}: frames sharded over ranks, no data-path collective"),
                           "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None},
                "roofline": {"bound": "mfma", "achieved": round(gflop * B / ms_per_step, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(gflop * B / ms_per_step / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None,
                             "kernel": "whole step (direct-algorithm FLOPs of the model over the step time; per-kernel tables: profiles/)"},
                "cpu_baseline": None}
        if train:
            line.update({"allreduce_us": round(ar_us, 1), "allreduce_ms": round(ar_us / 1e3, 4), "allreduce_bytes": nparam * 4,
                         "final_loss": float(last["loss"])})
        print(json.dumps(line), flush=True)
    parallel.barrier(dev)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process - which has not touched the GPU (importing torch
    and counting devices do not initialise it) - starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same
    arguments>` as a CHILD (never an exec), lets rank 0's JSON line through on stdout, and exits with the child's code.  Fewer GPUs than
    ranks is an error, not a silent one-GPU measurement.  Returns only when no launch is needed (N = 1, or already a rank)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    from eemflow_amd import parallel
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not parallel.share_gpu():
        print(f"bench.py: --gpus {args.gpus} but {n_dev} GPU(s) visible on this node: refusing to measure fewer GPUs than asked",
              file=sys.stderr, flush=True)
        sys.exit(2)
    with socket.socket() as sock:                              # a free rendezvous port on the loop-back interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's cross-process buffers on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    self_launch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with --nproc-per-node {args.gpus}",
              file=sys.stderr, flush=True)
        sys.exit(2)
    if args.rows_child:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
        torch.cuda.set_device(0)
        print(json.dumps(other_rows(torch.device("cuda", 0))), flush=True)
        return
    if args.workload == "eraft":
        return main_eraft(args)
    if args.mode == "train":
        return main_train(args)
    from eemflow_amd import _lib, parallel
    from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair

    rank, local_rank, world = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    dev = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    if world > 1:
        parallel.pin_host_threads_to_gpu_numa(dev.index)         # one rank per GPU: its host threads next to that GPU

    B, H, W = args.batch, args.height, args.width
    L = _lib.lib()
    sd_np = seeded_state_dict(0)
    e1_np, e2_np = synthetic_voxel_pair(1 + rank, B, H, W)          # each rank its own frames
    e1, e2 = torch.from_numpy(e1_np).to(dev), torch.from_numpy(e2_np).to(dev)
    # The timed loop walks ROTATE_PAIRS distinct input pairs in distinct buffers (pair 0 as generated, pair r = pair 0 rolled by r * (37, 53)
    # pixels: the same statistics, other bytes at other addresses): at 1280x720 that is 8 x 36.9 MB = 295 MB of inputs between two uses
    # of the same line, more than the 256 MiB Infinity Cache, so the first layer's reads come from HBM as they do in an evaluation loop
    n_rot = max(1, args.rotate_pairs)
    pairs = [(e1, e2)] + [(torch.roll(e1, shifts=(37 * r, 53 * r), dims=(2, 3)).contiguous(),
                           torch.roll(e2, shifts=(37 * r, 53 * r), dims=(2, 3)).contiguous()) for r in range(1, n_rot)]
    flat = torch.cat([torch.from_numpy(v).reshape(-1) for v in sd_np.values()]).contiguous()

    CO = max(1, args.coalesce) if B == 1 else 1          # frames per library call (batches > 1 are already batched chains)
    NS = args.streams if args.streams > 0 else (2 if CO > 1 else 4)
    ctxs, streams = [], []
    for _ in range(NS):
        c = ctypes.c_void_p()
        _lib.check(L.eemflow_create(dev.index, ctypes.byref(c)))
        _lib.check(L.eemflow_load_weights(c, flat.data_ptr(), flat.numel(), 5, 5))
        _lib.check(L.eemflow_set_image_size(c, H, W, None))
        _lib.check(L.eemflow_use_graph(c, 0 if args.no_graph else 1))
        _lib.check(L.eemflow_set_frames_in_flight(c, args.frames_in_flight or NS))
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=dev))
    # every frame in flight writes its own flow tensor: NS chains x CO frames, and one more set so that a chain's next call does not
    # rewrite tensors its previous call may still be writing... (same stream: ordered) - NS * CO distinct buffers suffice
    flows = [torch.empty(B, 2, H, W, device=dev) for _ in range(NS * CO)]
    ctx, stream, flow = ctxs[0], streams[0], flows[0]
    sp = ctypes.c_void_p(stream.cuda_stream)
    sps = [ctypes.c_void_p(st.cuda_stream) for st in streams]
    counter = [0]
    calls = [0]
    pending = []

    stagger = float(os.environ.get("EEM_BENCH_STAGGER_US", "0")) * 1e-6      # experiment: host pause between the first NS launches

    def flush():
        """Hand the frames collected so far to the library: ONE eemflow_forward_many call on the next chain."""
        n = len(pending)
        if n == 0:
            return
        i = calls[0] % NS
        calls[0] += 1
        arr = ctypes.c_void_p * n
        p1 = arr(*[f[0].data_ptr() for f in pending])
        p2 = arr(*[f[1].data_ptr() for f in pending])
        po = arr(*[f[2].data_ptr() for f in pending])
        _lib.check(L.eemflow_forward_many(ctxs[i], n, p1, p2, po, H, W, H, W, sps[i]))
        pending.clear()

    def step():
        """One frame (batch B = 1 pair) enters the pipeline."""
        if CO == 1:
            i = counter[0] % NS
            if stagger > 0 and 0 < counter[0] < NS:
                t_end = time.perf_counter() + stagger
                while time.perf_counter() < t_end:
                    pass
            a, b = pairs[counter[0] % n_rot]
            counter[0] += 1
            _lib.check(L.eemflow_forward(ctxs[i], a.data_ptr(), b.data_ptr(), B, H, W, flows[i].data_ptr(), H, W, sps[i]))
            return flows[i]
        a, b = pairs[counter[0] % n_rot]
        dst = flows[(calls[0] % NS) * CO + len(pending)]
        pending.append((a, b, dst))
        counter[0] += 1
        if len(pending) == CO:
            flush()
        return dst

    def timed(nsteps):
        """nsteps frames through the loop, bracketed as the contract says; returns (elapsed s, HIP-event ms, host enqueue s)."""
        torch.cuda.synchronize(dev)
        parallel.barrier(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        counter[0] = 0
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(nsteps):
            step()
        flush()                                          # a last, shorter call when CO does not divide nsteps
        enq = time.perf_counter() - t0                   # host time to hand the steps to the streams (reported; < elapsed = GPU-bound)
        for st in streams[1:]:
            stream.wait_stream(st)
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        parallel.barrier(dev)
        return el, ev0.elapsed_time(ev1), enq

    # Clock / runtime pre-heat (untimed, reported as "preheat_steps"): the contexts have just been built on an idle GPU, and the first
    # milliseconds after that run at ramping clocks with cold graph-launch paths (20 timed steps after 5 warm-up steps read 6 % lower than
    # after 100).  Every call size the loops below will issue (CO, and the remainders of the step counts) is captured here, on every
    # chain.  Then the W warm-up steps the caller asked for, then the timed region.
    for _ in range(args.preheat):
        step()
    flush()
    if CO > 1:
        for r in sorted({args.steps % CO, args.warmup % CO, args.long_steps % CO} - {0}):
            for _ in range(NS):
                for _ in range(r):
                    step()
                flush()
    torch.cuda.synchronize(dev)
    counter[0] = 0
    for _ in range(args.warmup):
        step()
    flush()
    elapsed, gpu_ms, enqueue_s = timed(args.steps)
    value, slowest = parallel.aggregate_throughput(args.steps * B, elapsed, dev)
    value_long = None
    if args.long_steps > 0:
        el_long, _, _ = timed(args.long_steps)
        value_long, _ = parallel.aggregate_throughput(args.long_steps * B, el_long, dev)

    if rank == 0:
        # ---- per-kernel roofline, measured live with HIP events on the launch stream, in the launch configuration of the timed loop
        # (frames_in_flight = streams: persistent encoder kernels on fewer blocks) and in the single-frame configuration (full grids)
        def kernel_table(frames_in_flight, nb=B, chain=True):
            # nb frames per launch: the timed loop's chains are batch-CO chains (the frames of one eemflow_forward_many call)
            _lib.check(L.eemflow_set_frames_in_flight(ctx, frames_in_flight))
            stats = (_lib.KernelStat * 64)()
            n = ctypes.c_int(0)
            if nb == B:
                k1, k2, kf = e1, e2, flow
            else:
                k1 = torch.cat([pairs[r % n_rot][0] for r in range(nb)]).contiguous()
                k2 = torch.cat([pairs[r % n_rot][1] for r in range(nb)]).contiguous()
                kf = torch.empty(nb, 2, H, W, device=dev)
            # chain: every launch timed in the chain behind its producer (the timed loop's batch-CO launches: the form rocprofv3's per-kernel
            # averages of the loop agree with); else each kernel repeated back to back (the single-frame table's few-microsecond launches)
            _lib.check(L.eemflow_time_kernels(ctx, k1.data_ptr(), k2.data_ptr(), nb, H, W, kf.data_ptr(), H, W,
                                              args.kernel_reps if chain else -args.kernel_reps, stats, 64, ctypes.byref(n), sp))
            torch.cuda.synchronize(dev)
            table = []
            for i in range(n.value):
                k = stats[i]
                sec = k.ms * 1e-3
                ai = k.flops / max(k.bytes, 1.0)
                # the roof a launch runs against: fp32 MFMA, or - for the kernels that compute fp32 products as six bf16-piece MFMAs -
                # the bf16 pipe / 6 (416.7 TFLOP/s, balance 52 FLOP/B: the stride-2 layers' 24-48 FLOP/B are then HBM-bound)
                # Winograd launches: `flops` are the DIRECT convolution's (SURVEY 8d's unit), the kernel multiplies a quarter (F(4x4,3x3))
                # or 1 / 2.25 (F(2x2,3x3)) of them - its matrix roof in that unit is 4 x / 2.25 x the fp32 MFMA peak, and with a balance of
                # 79 / 44 FLOP/B the 16- and 32-channel layers (32 - 64 FLOP/B) are HBM-bound
                peak_tf = {0: PEAK_MFMA_F32_TFLOPS, 1: PEAK_BF16X3_TFLOPS, 2: PEAK_MFMA_F32_TFLOPS * 4.0, 3: PEAK_MFMA_F32_TFLOPS * 2.25}.get(k.pipe, PEAK_MFMA_F32_TFLOPS)
                bound = "mfma" if ai >= peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
                # the encoder's kernels are persistent (one workgroup per CU): a launch of fewer than 256 workgroups occupies that many
                # CUs and leaves the rest to the other frames in flight; chip_us = duration x the share of the chip it holds
                cus = min(k.blocks, N_CUS) if k.blocks > 0 else N_CUS
                table.append({"name": k.name.decode(), "us": round(k.ms * 1e3, 2), "gflop": round(k.flops / 1e9, 4),
                              "mbytes": round(k.bytes / 1e6, 3), "bound": bound, "pipe": {0: "f32", 1: "bf16x3", 2: "f32 winograd F(4x4)", 3: "f32 winograd F(2x2)"}.get(k.pipe, "f32"),
                              "peak_tflops": round(peak_tf, 1),
                              "tflops": round(k.flops / sec / 1e12, 2), "gbs": round(k.bytes / sec / 1e9, 1),
                              "workgroups": k.blocks, "cus": cus, "chip_us": round(k.ms * 1e3 * cus / N_CUS, 2)})
            return table

        def roofline_of(table):
            # dominant kernel = the launch that holds the most of the chip for the longest (duration x CUs occupied); with every kernel
            # on the full chip this is the longest launch, as before.  achieved / frac are per launch against the WHOLE chip's peak (the
            # contract's definition); *_on_its_cus is the same against the peak of the CUs the launch occupies
            dom = max(table, key=lambda k: k["chip_us"])
            share = dom["cus"] / N_CUS
            if dom["bound"] == "mfma":
                r = {"bound": "mfma", "achieved": dom["tflops"], "peak": dom["peak_tflops"], "unit": "TFLOP/s",
                     "frac": round(dom["tflops"] / dom["peak_tflops"], 4)}
            else:
                r = {"bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(dom["gbs"] / PEAK_HBM_GBS, 4)}
            r["cus"] = dom["cus"]
            r["frac_on_its_cus"] = round(r["frac"] / share, 4) if r["bound"] == "mfma" else None
            tr_all = load_traffic() or {}
            wl = dict(tr_all.get("_workload", {"height": 720, "width": 1280, "batch": 1}))
            wl.setdefault("frames_per_launch", 1)
            same_shape = wl == {"height": H, "width": W, "batch": B, "frames_per_launch": CO * B}
            traffic = tr_all.get(dom["name"]) if same_shape else None        # PMC passes of another shape say nothing here
            # HBM bytes per launch of the dominant kernel from the PMC passes (profiles/pmc_traffic.json, tools/profile_gpu.sh): a
            # committed measurement, not a counter of this run - stamped with the kernel symbol and commit it was taken on, and
            # dropped when that symbol is not the one this library launches for the layer (tools/pmc_traffic.py writes both)
            meta = tr_all.get("_meta", {})
            r["traffic"] = traffic["hbm_bytes"] if isinstance(traffic, dict) and "hbm_bytes" in traffic else None
            r["traffic_detail"] = traffic
            r["traffic_source"] = ({"file": "profiles/pmc_traffic.json", "commit": meta.get("commit"), "date": meta.get("date"),
                                    "csrc_sha": meta.get("csrc_sha")} if traffic else
                                   ("stale: csrc/ changed since profiles/pmc_traffic.json was measured" if tr_all.get("_stale") else None))
            r["algorithmic_bytes"] = round(dom["mbytes"] * 1e6)
            r["kernel"] = dom["name"]
            r["kernel_us"] = dom["us"]
            r["pipe"] = dom["pipe"]
            # the same figures for the launch that takes the longest (dominant by TIME, whatever share of the chip it holds)
            lg = max(table, key=lambda k: k["us"])
            r["dominant_by_time"] = {"kernel": lg["name"], "kernel_us": lg["us"], "cus": lg["cus"], "bound": lg["bound"], "pipe": lg["pipe"],
                                     "achieved": lg["tflops"] if lg["bound"] == "mfma" else lg["gbs"],
                                     "peak": lg["peak_tflops"] if lg["bound"] == "mfma" else PEAK_HBM_GBS,
                                     "unit": "TFLOP/s" if lg["bound"] == "mfma" else "GB/s",
                                     "frac": round(lg["tflops"] / lg["peak_tflops"], 4) if lg["bound"] == "mfma" else round(lg["gbs"] / PEAK_HBM_GBS, 4)}
            r["longest_launch"] = {k2: lg[k2] for k2 in ("name", "us", "cus", "tflops")}
            return r

        fif = args.frames_in_flight or NS
        kernels = kernel_table(fif, CO * B)
        roof = roofline_of(kernels)
        roof["frames_in_flight"] = fif
        roof["frames_per_launch"] = CO * B
        roof_single = None
        if not args.no_side_rows:
            single = kernel_table(1, chain=False)
            roof_single = roofline_of(single)
            roof_single["frames_in_flight"] = 1
            roof_single["kernels_us"] = {k["name"].split()[0]: k["us"] for k in single}
            _lib.check(L.eemflow_set_frames_in_flight(ctx, fif))
        sum_us = sum(k["us"] for k in kernels)
        enc = [k for k in kernels if k["name"].startswith("enc.")]
        enc_tflops = sum(k["gflop"] for k in enc) / max(sum(k["us"] for k in enc), 1e-9) * 1e3
        total_gflop = sum(k["gflop"] for k in kernels) / CO        # of one STEP (one batch-B forward): the table's launches carry CO of them

        # ---- CPU baseline + EPE agreement on this rank's frames
        cpu = None
        extra = {}
        if args.cpu_seconds > 0 and world == 1:              # the CPU baseline is timed at N=1 only
            from oracle import eemflow_oracle as O
            ref, cpu = cpu_baseline(sd_np, torch.from_numpy(e1_np), torch.from_numpy(e2_np), args.cpu_seconds)
            # Parity of the TIMED launch configuration: one full call of the timed loop - CO frames through ONE
            # eemflow_forward_many call on chain 0 (batch-CO kernels: F(4x4) everywhere, column walk, multi-tile decoder) - and every frame
            # of it against an oracle forward of ITS input pair (the rotated pairs are rolled copies of pair 0: min(CO, n_rot) distinct
            # oracle forwards; the oracle is the checker here, after the timed region).
            counter[0] = 0
            calls[0] = 0
            dsts = [step() for _ in range(CO)]               # the CO-th step flushes: one call
            flush()
            torch.cuda.synchronize(dev)
            refs = {0: ref}
            sd_t = O.to_torch_sd(sd_np)
            errs = []
            with torch.no_grad():
                for k, d in enumerate(dsts):
                    r = k % n_rot
                    if r not in refs:
                        refs[r] = O.eemflow_forward(sd_t, pairs[r][0].cpu(), pairs[r][1].cpu())[0]
                    errs.append(float((d.cpu() - refs[r]).abs().max()))
            got = dsts[0].cpu()
            yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
            gt = np.stack([3 * np.sin(2 * np.pi * xx / W), 3 * np.cos(2 * np.pi * yy / H)])
            extra = {"flow_max_abs_err_vs_oracle": max(errs),
                     "flow_max_abs_err_vs_oracle_frames": len(errs),
                     "flow_max_abs_err_vs_oracle_note": f"max over the {len(errs)} frames of ONE timed-loop call (eemflow_forward_many, "
                                                        f"{CO} frames per launch; {len(refs)} distinct input pairs, one oracle forward each)",
                     "epe_hip": O.flow_error_dense(gt, got[0].numpy())[0],
                     "epe_oracle": O.flow_error_dense(gt, ref[0].numpy())[0]}
            extra["speedup_vs_cpu_baseline"] = round(value / world / cpu["value"], 1)
            # BASELINE configs[0]: the reference's own CPU-runnable case, one synthetic 346x260x5 pair (bounded sample)
            s1, s2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(1, 1, 260, 346))
            _, cpu_small = cpu_baseline(sd_np, s1, s2, min(3.0, args.cpu_seconds))
            extra["cpu_baseline_346x260"] = cpu_small

        ms_per_step = slowest * 1e3 / args.steps
        # the robust number: the frame's direct-convolution FLOPs at the fp32 MFMA peak (its ceiling with the direct algorithm on that
        # pipe) over the measured time per step of one GPU
        ceiling_ms = total_gflop / PEAK_MFMA_F32_TFLOPS                    # GFLOP of one step (its whole batch) / (TFLOP/s) = ms
        roof["frame_frac_of_ceiling"] = round(ceiling_ms / ms_per_step, 4)
        roof["frame_ceiling_us"] = round(ceiling_ms * 1e3, 2)
        # ... and the frame against the HBM roof: the encoder's measured HBM-side bytes per frame (the committed PMC passes, same sources
        # and launch shape) over the measured time per frame - the first three layers run at their bytes, so this is the bound that binds
        tr_now = load_traffic() or {}
        if tr_now.get("_encoder_mb_per_frame") and not tr_now.get("_stale") and (tr_now.get("_workload") or {}).get("frames_per_launch", 1) == CO * B:
            roof["frame_hbm_mbytes"] = tr_now["_encoder_mb_per_frame"]
            roof["frame_hbm_gbs"] = round(tr_now["_encoder_mb_per_frame"] / 1e3 / (ms_per_step * 1e-3), 1)
            roof["frame_hbm_frac"] = round(roof["frame_hbm_gbs"] / PEAK_HBM_GBS, 4)
        line = {
            "metric": baseline_metric(),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"EEMFlow inference, HREM {W}x{H} dt1, batch={B} per GPU (BASELINE configs[1]); "
                                   "synthetic 20%-dense voxel pairs resident in HBM, seeded Kaiming weights",
                       "height": H, "width": W, "batch_per_gpu": B, "hip_graph": not args.no_graph, "streams_per_gpu": NS,
                       "coalesced_frames": CO, "frames_in_flight_hint": args.frames_in_flight or NS, "preheat_steps": args.preheat,
                       "input_pairs_rotated": n_rot, "input_bytes_rotated": n_rot * 2 * e1.numel() * 4,
                       "parallelism": f"replicas x{world}: frames sharded over ranks, no data-path collective"},
            "roofline": roof, "roofline_single_frame_launch": roof_single, "cpu_baseline": cpu,
            "value_long": None if value_long is None else round(value_long, 2), "value_long_steps": args.long_steps,
            "gpu_ms_per_step_hip_events": round(gpu_ms / args.steps, 4),
            "host_enqueue_ms_per_step": round(enqueue_s * 1e3 / args.steps, 4),
            "schedule_sum_us": round(sum_us, 1), "schedule_frames_per_launch": CO * B, "frame_gflop": round(total_gflop, 3),
            "frame_tflops": round(total_gflop / ms_per_step, 2),
            "encoder_tflops_in_kernel": round(enc_tflops, 2),
            "kernels": kernels, **extra,
        }
        # The other rows are measured in a FRESH child process (its own HIP context; this one idles meanwhile): in this process, behind the
        # batch-10 workspaces of the main loop, whichever of them comes late runs slower ON THE GPU with the host enqueue unchanged -
        # single-stream E-RAFT 178 -> 70 frames/s behind the pipeline section, or the E-RAFT training step 94 -> 112 ms when the rows come
        # first (measured, not explained: docs/NOTEBOOK.md section 10).  A child process is what `tools/bench_eraft*.py` are, and agrees with them.
        if not args.no_other_rows and world == 1:
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rows-child"], capture_output=True, text=True, timeout=900)
                line["other_rows"] = json.loads(r.stdout.strip().splitlines()[-1])
                line["other_rows"]["measured_in"] = "child process (fresh HIP context)"
            except Exception as e:                               # noqa: BLE001 - the side rows are never fatal
                line["other_rows"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_side_rows:
            rows = latency_and_pipeline_rows(L, _lib, dev, flat, B, H, W, e1, e2, not args.no_graph, CO)
            line["latency_ms_b1"] = rows.pop("latency_ms_b1")
            line["latency_and_pipeline"] = rows
        print(json.dumps(line), flush=True)
    parallel.barrier(dev)
    for c in ctxs:
        L.eemflow_destroy(c)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
