"""GPU box: E-RAFT's parameter gradient (test_eraft_loss_backward_vs_oracle_autograd's case) against the oracle run in float64 - for the
float32 oracle, and for this library with the stride-2 convs on gconv16 and on the generic kernel.  ReLU / instance-norm chains over
16 x 20 maps are chaotic in fp32: what counts is whether the library is as close to the fp64 gradient as the fp32 oracle is."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_eraft_train import make_model, DEV, R, T
from eemflow_amd import train as hip_train
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
b, h, w, iters = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 128, 160, 3)))


def oracle(sd, e1, e2, gt, valid, dt):
    params = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and "running_" not in k else (v.clone().to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    keep, R.FLOAT = R.FLOAT, dt                                   # the oracle's `.float()` casts (the reference's) go through this name
    try:
        preds, _ = R.eraft_forward(params, e1.to(dt), e2.to(dt), iters=iters, image_size=(h, w), bn_training=True)
    finally:
        R.FLOAT = keep
    loss, _ = T.sequence_loss(preds, gt.to(dt), valid.to(dt), 0.8)
    loss.backward()
    return {k: v.grad.double() for k, v in params.items() if v.is_floating_point() and v.requires_grad and v.grad is not None}


def hip(flag):
    os.environ["EEM_NO_G16_S2"] = flag
    net, sd = make_model(31)
    net.change_imagesize((h, w))
    (_, _), preds = net(e1.to(DEV), e2.to(DEV), iters=iters)
    loss, _ = hip_train.sequence_loss(preds, gt.to(DEV), valid.to(DEV), 0.8)
    loss.backward()
    return {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}, sd


e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(32, b, h, w))
gt, valid = (torch.from_numpy(a) for a in synthetic_gt(33, b, h, w))
g_on, sd = hip("0")
g_off, _ = hip("1")
g64 = oracle(sd, e1, e2, gt, valid, torch.float64)
g32 = oracle(sd, e1, e2, gt, valid, torch.float32)
gmax = max(float(g.abs().max()) for g in g64.values())
live = [k for k, g in g64.items() if float(g.abs().max()) > 1e-6 * gmax]
for name, g in (("fp32 oracle", g32), ("hip, stride-2 on gconv16", g_on), ("hip, stride-2 generic", g_off)):
    errs = sorted(((float((g[k] - g64[k]).abs().max() / (g64[k].abs().max() + 1e-30)), k) for k in live), reverse=True)
    num = sum(float(((g[k] - g64[k]) ** 2).sum()) for k in live) ** 0.5
    den = sum(float((g64[k] ** 2).sum()) for k in live) ** 0.5
    print("%-28s vs fp64: rel L2 %.3e, tensors >= 5e-3: %d / %d, worst %.3e %s" % (name, num / den, sum(e >= 5e-3 for e, _ in errs), len(errs), errs[0][0], errs[0][1]))
