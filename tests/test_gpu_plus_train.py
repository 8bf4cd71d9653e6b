"""EEMFlow+ (EEMFlow_cdc) under autograd on the GPU (model/EEMFlow/EEMFlow+.py:158-234, cdc_utils.py:50-177): its operators against torch
autograd of the same op, the encoder + level 6 against the oracle, and every finer level TEACHER-FORCED - the same flow_init on both sides,
so that WarpingLayer_no_div's discontinuous `>= 1` mask is identical and the level's outputs and gradients can be held to a tolerance
(the chained model can only be compared statistically, see test_gpu_plus.py).  `pytest -m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from eemflow_amd import ops
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_gt, synthetic_voxel_pair
from oracle import eemflow_oracle as O
from oracle import eemflow_plus_oracle as P
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).float()


def make_model(seed, cin=5):
    net = EEMFlow_cdc("", 3, cin)
    sdn = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()})
    return net.to(DEV).train(), O.to_torch_sd(sdn)


def test_small_operators_vs_torch():
    x = rnd(2, 2, 5, 7, seed=1, scale=3.0)
    xr = x.clone().requires_grad_(True)
    inp = xr * 1.0
    res_r = P.upsample2d_flow_as(inp, (10, 13), if_rate=True)           # scales `inp` in place
    g1, g2 = rnd(*res_r.shape, seed=2), rnd(*x.shape, seed=3)
    ((res_r * g1).sum() + (inp * g2).sum()).backward()
    xg = x.to(DEV).requires_grad_(True)
    res_g, scaled_g = ops.UpsampleFlowAs.apply(xg, 10, 13)
    ((res_g * g1.to(DEV)).sum() + (scaled_g * g2.to(DEV)).sum()).backward()
    assert rel(res_g, res_r) < 1e-6 and rel(scaled_g, inp) < 1e-6 and rel(xg.grad, xr.grad) < 1e-5
    # pooling (odd sizes), shuffle, slices, cat, sigmoid
    y = rnd(2, 6, 7, 9, seed=4)
    yr = y.clone().requires_grad_(True)
    ref = F.avg_pool2d(yr, 2, 2)
    sh_r = O.channel_shuffle(yr, 3)
    cat_r = torch.cat([torch.sigmoid(yr[:, 4:5]), yr[:, :2]], 1)
    gp, gs, gc = rnd(*ref.shape, seed=5), rnd(*y.shape, seed=6), rnd(*cat_r.shape, seed=7)
    ((ref * gp).sum() + (sh_r * gs).sum() + (cat_r * gc).sum()).backward()
    yg = y.to(DEV).requires_grad_(True)
    pool_g, sh_g = ops.AvgPool2.apply(yg), ops.ChannelShuffle.apply(yg, 3)
    cat_g = ops.CatN.apply(ops.Sigmoid.apply(ops.ChannelSlice.apply(yg, 4, 1)), ops.ChannelSlice.apply(yg, 0, 2))
    ((pool_g * gp.to(DEV)).sum() + (sh_g * gs.to(DEV)).sum() + (cat_g * gc.to(DEV)).sum()).backward()
    assert rel(pool_g, ref) < 1e-6 and rel(sh_g, sh_r) == 0.0 and rel(cat_g, cat_r) < 1e-6 and rel(yg.grad, yr.grad) < 1e-5
    # 53-tap local correlation
    a, b = rnd(2, 8, 9, 11, seed=8), rnd(2, 8, 9, 11, seed=9)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    cr = P.corr53(ar, br)
    gk = rnd(*cr.shape, seed=10)
    cr.backward(gk)
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    cg = ops.LocalCorr53.apply(ag, bg)
    cg.backward(gk.to(DEV))
    assert rel(cg, cr) < 1e-5 and rel(ag.grad, ar.grad) < 1e-5 and rel(bg.grad, br.grad) < 1e-5


def test_decoder_with_grouped_convs_and_shuffle_vs_oracle():
    net, sd = make_model(3)
    x = rnd(2, 87, 8, 12, seed=11)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder4.")}
    xr = x.clone().requires_grad_(True)
    out_r = P.decoder(params, "decoder4.", xr, 3)
    g = rnd(*out_r.shape, seed=12)
    out_r.backward(g)
    xg = x.to(DEV).requires_grad_(True)
    out_g = net._decoder_ops(net.decoder4, xg)
    out_g.backward(g.to(DEV))
    assert rel(out_g, out_r) < 1e-4 and rel(xg.grad, xr.grad) < 1e-3
    named = dict(net.named_parameters())
    worst = max((rel(named[k].grad, v.grad), k) for k, v in params.items())
    assert worst[0] < 3e-3, worst


def test_encoder_and_level6_gradients_vs_oracle():
    """Loss on the coarsest prediction only: pad, shared encoder, poolings, correlation, rconv6, decoder6 - no warp upstream."""
    b, h, w = 2, 128, 192
    net, sd = make_model(5)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(6, b, h, w, bins=5))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(7, b, h, w))
    preds = net(e1.to(DEV), e2.to(DEV))[1]
    loss, _ = T.sequence_loss(preds[:1], gt.to(DEV), valid.to(DEV))
    loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rpreds, _ = P.eemflow_plus_forward(params, e1, e2)
    rloss, _ = T.sequence_loss(rpreds[:1], gt, valid)
    rloss.backward()
    assert abs(float(loss) - float(rloss)) < 1e-5 and float((preds[0].detach().cpu() - rpreds[0].detach()).abs().max()) < 1e-3
    named = dict(net.named_parameters())
    live = {k: v.grad for k, v in params.items() if v.grad is not None and float(v.grad.abs().max()) > 0}
    assert len(live) >= 8 * 2 + 2 + 14
    worst = max((rel(named[k].grad, g), k) for k, g in live.items())
    assert worst[0] < 5e-3, worst


@pytest.mark.parametrize("l", [5, 4, 3, 2])
def test_levels_teacher_forced_outputs_and_gradients_vs_oracle(l):
    b, h, w = 2, 128, 192
    net, sd = make_model(9)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(10, b, h, w, bins=5))
    with torch.no_grad():
        _, st = P.eemflow_plus_forward(sd, e1, e2, keep=True)
    f1l, f2l, init = st["f1"][l].contiguous(), st["f2"][l].contiguous(), st[f"flow_init{l}"].contiguous()
    keys = [k for k in sd if k.startswith((f"conv_1x1.{l}.", "cdc_model.dense_estimator_mask.", f"rconv{l}.", f"decoder{l}."))]
    params = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    ar, br, ir = (t.clone().requires_grad_(True) for t in (f1l, f2l, init))
    up_r, fl_r = P.level_from_init(params, l, ar, br, ir)
    g1, g2 = rnd(*up_r.shape, seed=20 + l), rnd(*fl_r.shape, seed=30 + l)
    ((up_r * g1).sum() + (fl_r * g2).sum()).backward()
    ag, bg, ig = (t.to(DEV).requires_grad_(True) for t in (f1l, f2l, init))
    up_g, fl_g = net._level_ops(l, ag, bg, ig)
    ((up_g * g1.to(DEV)).sum() + (fl_g * g2.to(DEV)).sum()).backward()
    assert float((up_g.detach().cpu() - up_r.detach()).abs().max()) < 1e-3 and float((fl_g.detach().cpu() - fl_r.detach()).abs().max()) < 1e-3
    for x, y, nm in ((ag, ar, "f1"), (bg, br, "f2"), (ig, ir, "flow_init")):
        assert rel(x.grad, y.grad) < 5e-3, (nm, rel(x.grad, y.grad))
    named = dict(net.named_parameters())
    worst = max((rel(named[k].grad, params[k].grad), k) for k in keys if params[k].grad is not None and float(params[k].grad.abs().max()) > 0)
    assert worst[0] < 5e-3, worst


def test_reference_training_sequence_on_the_chained_model():
    """train_mvsec.py:241-258 on EEMFlow_cdc: five predictions, gamma-weighted loss, torch AdamW; the loss falls, and the inference
    route then agrees with the stepped module's own autograd-route forward at the coarsest level."""
    from eemflow_amd import train as hip_train
    b, h, w = 1, 128, 192
    model, _ = make_model(13)
    model.change_imagesize((h, w))
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=2e-4, weight_decay=5e-5, eps=1e-8)
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(14, b, h, w, bins=5))
    gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(15, b, h, w))
    losses = []
    for _ in range(4):
        optimizer.zero_grad()
        _, flow_list = model(e1, e2)
        assert len(flow_list) == 5 and flow_list[-1].shape == (b, 2, h, w)
        loss, _ = hip_train.sequence_loss(flow_list, gt, valid, 0.8)
        loss.backward()
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    used = [k for k, p in model.named_parameters() if p.grad is not None]
    assert len(used) >= 100                                       # the parameters the reference never uses (up3..6, ...) get none
    model.eval()
    with torch.no_grad():
        inf = model(e1, e2)[1]
    ops_preds = model._forward_ops(e1, e2)
    assert float((inf[0] - ops_preds[0].detach()).abs().max()) < 1e-3
