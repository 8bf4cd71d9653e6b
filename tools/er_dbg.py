import sys, os
sys.path.insert(0, os.getcwd())
import torch
from eemflow_amd.eraft import ERAFT
from eemflow_amd.eraft_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair
h, w = int(sys.argv[1]), int(sys.argv[2])
net = ERAFT("", 5).eval()
sd = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda(); net.change_imagesize((h, w))
e1, e2 = (torch.from_numpy(a).cuda() for a in synthetic_voxel_pair(1, 1, h, w))
with torch.no_grad():
    out = net(e1, e2, iters=3)[1]
    torch.cuda.synchronize()
print("ok", float(out[-1].abs().max()))
