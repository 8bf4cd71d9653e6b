#!/usr/bin/env python3
"""Soak run of what round 4 added, on the GPU box: (1) E-RAFT forwards with the side-stream branches on three contexts / streams at once,
every result compared bitwise with the one-stream schedule's (EEM_ERAFT_NO_OVERLAP is read per forward); (2) voxelizer pair calls on four
streams with different event sets, compared with the same calls run alone; (3) EEMFlow inference on four streams with rotating inputs
against the results of a single stream.  Nothing may differ, hang or grow."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                # noqa: E402
import torch                                                      # noqa: E402
from eemflow_amd import EEMFlow                                   # noqa: E402
from eemflow_amd.eraft import ERAFT                               # noqa: E402
from eemflow_amd.eraft_weights import seeded_from_shapes          # noqa: E402
from eemflow_amd.hrem import synthetic_hrem_events                # noqa: E402
from eemflow_amd.voxelizer import EventSequence, voxelize_pair_device   # noqa: E402
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair  # noqa: E402

dev = torch.device("cuda:0")


def used():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 2**20


t_all = time.perf_counter()
# ---- (1) E-RAFT, side stream against one stream
nets = []
for _ in range(3):
    net = ERAFT("", 5).eval()
    sd = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev); net.change_imagesize((480, 640)); nets.append(net)
streams = [torch.cuda.Stream() for _ in nets]
pairs = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(10 + i, 1, 480, 640)) for i in range(5)]
with torch.no_grad():
    os.environ["EEM_ERAFT_NO_OVERLAP"] = "1"
    refs = [nets[0](p[0], p[1], iters=12)[1][-1].clone() for p in pairs]
    torch.cuda.synchronize()                                     # (a context serves one stream at a time: the reference runs are done before it moves on)
    os.environ["EEM_ERAFT_NO_OVERLAP"] = "0"
    for k in range(3):                                           # every context's workspace and side stream exist before the memory reading
        with torch.cuda.stream(streams[k]):
            for p in pairs: nets[k](p[0], p[1], iters=12)
    torch.cuda.synchronize(); m0 = used(); bad = 0
    n = 600
    outs = []
    for i in range(n):
        k = i % 3
        with torch.cuda.stream(streams[k]):
            outs.append((i % 5, nets[k](*pairs[i % 5], iters=12)[1][-1]))
        if len(outs) >= 30:
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, refs[j]) else 1 for j, o in outs)
            outs = []
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(o, refs[j]) else 1 for j, o in outs)
print("E-RAFT side stream: %d forwards on 3 streams, %d differ from the one-stream schedule, memory %+.1f MiB (the outputs held for comparison, cached by torch)" % (n, bad, used() - m0), flush=True)
assert bad == 0
del nets, refs, outs

# ---- (2) voxelizer pairs on four streams
H, W = 720, 1280
sets = []
for i in range(6):
    evs = []
    for j in range(2):
        ev = synthetic_hrem_events(100 + 2 * i + j, 150000 + 40000 * i, H, W)
        seq = EventSequence(None, {"height": H, "width": W}, features=ev, timestamp_multiplier=1e6, convert_to_relative=True)
        evs.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(dev))
    sets.append(evs)
refs = [tuple(g.clone() for g in voxelize_pair_device(s[0], s[1], 5, H, W, True)) for s in sets]
torch.cuda.synchronize()
vstreams = [torch.cuda.Stream() for _ in range(4)]
for i in range(48):
    with torch.cuda.stream(vstreams[i % 4]):
        voxelize_pair_device(sets[i % 6][0], sets[i % 6][1], 5, H, W, True)
torch.cuda.synchronize(); m0 = used(); bad = 0; worst = 0.0
outs = []
n = 2000
for i in range(n):
    with torch.cuda.stream(vstreams[i % 4]):
        outs.append((i % 6, voxelize_pair_device(sets[i % 6][0], sets[i % 6][1], 5, H, W, True)))
    if len(outs) >= 40:
        torch.cuda.synchronize()
        for j, (a, b) in outs:
            d = max(float((a - refs[j][0]).abs().max()), float((b - refs[j][1]).abs().max()))
            worst = max(worst, d); bad += d > 2e-5
        outs = []
torch.cuda.synchronize()
print("voxelizer pairs: %d calls on 4 streams, worst difference to the call alone %.2e, %d beyond 2e-5, memory %+.1f MiB (the outputs held for comparison, cached by torch)" % (n, worst, bad, used() - m0), flush=True)
assert bad == 0

# ---- (3) EEMFlow on four streams, rotating inputs
enets = []
for _ in range(4):
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
    net = net.to(dev).eval(); net.change_imagesize((720, 1280)); net.frames_in_flight = 4; enets.append(net)
pairs = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(30 + i, 1, 720, 1280)) for i in range(8)]
with torch.no_grad():
    refs = [enets[0](*p)[1][0].clone() for p in pairs]
    torch.cuda.synchronize()
    estreams = [torch.cuda.Stream() for _ in enets]
    for i in range(64):
        with torch.cuda.stream(estreams[i % 4]):
            enets[i % 4](*pairs[i % 8])
    torch.cuda.synchronize(); m0 = used(); bad = 0
    outs = []
    n = 20000
    for i in range(n):
        with torch.cuda.stream(estreams[i % 4]):
            outs.append((i % 8, enets[i % 4](*pairs[i % 8])[1][0]))
        if len(outs) >= 64:
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, refs[j]) else 1 for j, o in outs)
            outs = []
    torch.cuda.synchronize()
print("EEMFlow: %d frames on 4 streams with 8 rotating input pairs, %d differ from a single stream's, memory %+.1f MiB (the outputs held for comparison, cached by torch)" % (n, bad, used() - m0), flush=True)
assert bad == 0
print("soak ok in %.0f s" % (time.perf_counter() - t_all))
