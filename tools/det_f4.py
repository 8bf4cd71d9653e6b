import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from eemflow_amd import EEMFlow, _lib
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair, synthetic_gt
from eemflow_amd.train import EEMFlowTrainer
DEV = "cuda:0"
b, h, w = 4, 260, 346
e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(131, b, h, w))
gt, va = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(132, b, h, w))
net = EEMFlow("", 5, 5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(133).items()})
net = net.to(DEV).train(); net.change_imagesize((h, w))
ctx = net._context(torch.device(DEV))
L = _lib.lib()
n = sum(p.numel() for p in net.parameters())
sp = _lib.current_stream_ptr(torch.device(DEV))
ref = None; refg = None; bad = 0; badg = 0
keys = list(net.state_dict().keys()); sizes = [v.numel() for v in net.state_dict().values()]
for it in range(60):
    grad = torch.empty(n, device=DEV); flow = torch.empty(b, 2, h, w, device=DEV)
    stats = (ctypes.c_double * 5)()
    _lib.check(L.eemflow_forward_backward(ctx, e1.data_ptr(), e2.data_ptr(), gt.data_ptr(), va.data_ptr(), b, h, w, h, w, 1.0, flow.data_ptr(), grad.data_ptr(), ctypes.byref(stats), sp))
    torch.cuda.synchronize()
    f = (net.stage("f11").clone(), flow.clone())
    if ref is None: ref, refg = f, grad.clone()
    else:
        if not (torch.equal(ref[0], f[0]) and torch.equal(ref[1], f[1])): bad += 1
        d = (grad - refg).abs()
        rel = float(d.max() / refg.abs().max())
        if rel > 1e-5:
            badg += 1
            off = 0
            for k, sz in zip(keys, sizes):
                dd = float(d[off:off+sz].max()); 
                if dd > 1e-5 * float(refg.abs().max()): print("  it", it, k, dd, float(refg[off:off+sz].abs().max()))
                off += sz
print("forward mismatches", bad, "gradient outliers", badg, "of 59")
