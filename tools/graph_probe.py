"""GPU box: what a captured HIP graph of a whole forward would buy (torch.cuda.graph around the ctypes launches): EEMFlow+ and E-RAFT."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eemflow_amd.weights import synthetic_voxel_pair

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

def probe(name, net, e1, e2, **kw):
    with torch.no_grad():
        eager = timeit(lambda: net(e1, e2, **kw))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): net(e1, e2, **kw)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net(e1, e2, **kw)
        graph = timeit(g.replay)
    b = e1.shape[0]
    print("%s: eager %.3f ms (%.1f frames/s), graph replay %.3f ms (%.1f frames/s)" % (name, eager * 1e3, b / eager, graph * 1e3, b / graph), flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "plus"
if which == "plus":
    from eemflow_amd.eemflow_plus import EEMFlow_cdc
    from eemflow_amd.plus_weights import seeded_from_shapes
    net = EEMFlow_cdc("", 3, 5).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0).items()})
    net = net.cuda(); net.change_imagesize((720, 1280))
    e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, 1, 720, 1280))
    probe("EEMFlow+ 1280x720 b1", net, e1, e2)
else:
    from eemflow_amd.eraft import ERAFT
    from eemflow_amd.eraft_weights import seeded_from_shapes
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    net = ERAFT("", 5).eval()
    sd = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda(); net.change_imagesize((480, 640))
    e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, b, 480, 640))
    probe("E-RAFT 640x480 x12 b%d" % b, net, e1, e2, iters=12)
