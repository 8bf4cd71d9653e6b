import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_eraft_train import make_model, DEV
from eemflow_amd import train as hip_train
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
b, h, w, iters = 2, 128, 160, 3
res = {}
for mode in ("1", "0"):
    os.environ["EEM_NO_GCONV16"] = "1" if os.environ.get("PERTURB") else mode
    net, sd = make_model(31)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(32, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(33, b, h, w))
    if os.environ.get("PERTURB") and mode == "0":
        e1 = e1 * (1 + 1e-6 * torch.randn(e1.shape))
    (_, _), preds = net(e1.to(DEV), e2.to(DEV), iters=iters)
    loss, metrics = hip_train.sequence_loss(preds, gt.to(DEV), valid.to(DEV), 0.8)
    loss.backward()
    res[mode] = ({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, [p.detach().clone() for p in preds], float(loss))
print("loss", res["1"][2], res["0"][2], "pred diff", max(float((a - c).abs().max()) for a, c in zip(res["1"][1], res["0"][1])))
rows = []
gmax = max(float(v.abs().max()) for v in res["1"][0].values())
for k in res["1"][0]:
    a, c = res["1"][0][k], res["0"][0][k]
    if float(a.abs().max()) < 1e-6 * gmax:
        continue
    rows.append((float((a - c).abs().max() / (a.abs().max() + 1e-12)), k, tuple(a.shape)))
for r in sorted(rows, reverse=True)[:25]: print("%.3e %s %s" % r)
