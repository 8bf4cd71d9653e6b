import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair
from oracle import eemflow_oracle as O
from oracle import eemflow_plus_oracle as P
b, h, w, cin = 1, 720, 1280, 5
net = EEMFlow_cdc("", 3, cin).eval()
sdn = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 91)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()})
net = net.to("cuda:0"); sd = O.to_torch_sd(sdn); net.change_imagesize((h, w))
e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(92, b, h, w, bins=cin))
with torch.no_grad():
    net(e1.cuda(), e2.cuda())
    _, st = P.eemflow_plus_forward(sd, e1, e2, keep=True)
    out = []
    for l in (5, 4, 3, 2):
        init = net.stage(f"flow_init{l}")
        up_gpu = net.stage(f"flow_up{l}")
        up2, fl_gpu = net.level(l, init)
        up_ref, fl_ref = P.level_from_init(sd, l, st["f1"][l], st["f2"][l], init.cpu())
        d = (up_gpu.cpu() - up_ref).abs()
        out.append((l, float(d.max()), float((d > 1e-3).float().mean()), float((fl_gpu.cpu() - fl_ref).abs().max()), float(up_ref.abs().max())))
print(os.environ.get("TAG", ""), out)
