import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_train import make_net, run_grads, split_flat, T
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
b,h,w,size = 1,128,192,(128,192)
net, sd = make_net(23); net.change_imagesize(size)
e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(24, b, h, w))
gt, valid = (torch.from_numpy(a) for a in synthetic_gt(25, b, h, w))
loss, _, flow, flat = run_grads(net, e1, e2, gt, valid)
rloss, _, rg, rflow = T.loss_and_grads(sd, e1, e2, gt, valid, image_size=size)
g = split_flat(flat, sd)
rows = []
for k in sd:
    a, r = g[k].double(), rg[k].double()
    d = (a - r).abs(); m = r.abs().max() + 1e-12
    rows.append((float(d.max() / m), float((a - r).norm() / (r.norm() + 1e-12)), int((d > 1e-3 * m).sum()), a.numel(), k))
for r in sorted(rows, reverse=True)[:8]: print("max %.2e  l2 %.2e  n>1e-3 %d / %d  %s" % r)
