#!/usr/bin/env python3
"""Evaluation loop (harness.TestRaftEvents, files -> GPU voxelizer -> EEMFlow -> flow_error) on a synthetic HREM tree, one sample at
a time and with samples in flight: tools/eval_throughput.py [samples] [events per volume]"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                      # noqa: E402
from eemflow_amd import EEMFlow, hrem                             # noqa: E402
from eemflow_amd.harness import TestRaftEvents                    # noqa: E402
from eemflow_amd.weights import seeded_state_dict                 # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
root = tempfile.mkdtemp(prefix="hrem_")
try:
    for i in range(n):
        d = os.path.join(root, "dataset/HREM/test/dt1/seq/%06d" % (i + 1))
        os.makedirs(d)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(2 * i, nev, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(2 * i + 1, nev, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(i, 720, 1280))
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()})
    net = net.cuda()
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    for nfl, thr in ((1, 0), (4, 0), (1, 4), (4, 4), (4, 8), (1, 0)):
        ev = TestRaftEvents(hrem.HREMEventFlow(args, train=False, root=root), (720, 1280))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            aee = ev.test_multi_sequence(net, sequence_list=["seq"], stride=1, frames_in_flight=nfl, loader_threads=thr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"frames_in_flight={nfl} loader_threads={thr}: {n / dt:8.1f} samples/s ({dt / n * 1e3:.2f} ms per sample), mean AEE {aee:.6f}",
              flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
