"""The reference's own training call path through the HIP library (train_mvsec.py:241-258,377-386):
`model.train(); _, preds = model(im1, im2); loss, _ = sequence_loss(preds, gt, valid); scaler.scale(loss).backward();
clip_grad_norm_; scaler.step(optimizer); scheduler.step()` with torch's AdamW / OneCycleLR / GradScaler - nn.Parameter.grad
is filled by the model's torch.autograd.Function (eemflow_forward_train / eemflow_backward).  Checked against the golden
produced by the reference module and the reference's sequence_loss, and against the oracle.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from eemflow_amd import EEMFlow
from eemflow_amd import train as hip_train
from eemflow_amd.train import EEMFlowTrainer
from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
from oracle import eemflow_oracle as O
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_net(seed, **kw):
    sd = seeded_state_dict(seed)
    net = EEMFlow("", groups=5, n_first_channels=5, **kw)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(DEV).train(), O.to_torch_sd(sd)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def batch_of(seed_in, seed_gt, b, h, w, oh=None, ow=None):
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(seed_in, b, h, w))
    gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(seed_gt, b, oh or h, ow or w))
    return e1, e2, gt, valid


def test_loss_backward_fills_parameter_grads_vs_golden(golden):
    """model(e1, e2) -> the ORACLE's sequence_loss (plain torch ops on the HIP prediction) -> loss.backward()."""
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b = int(g["batch"])
    net, sd = make_net(int(g["seed"]))
    net.change_imagesize((h, w))
    e1, e2, gt, valid = batch_of(int(g["input_seed"]), int(g["gt_seed"]), b, h, w)
    (o1, o2), preds = net(e1, e2)
    assert o1 is e1 and o2 is e2 and len(preds) == 1 and preds[0].requires_grad
    loss, metrics = T.sequence_loss(preds, gt, valid, 0.8)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 and abs(metrics["epe"] - float(g["epe"])) < 1e-4
    assert float((preds[0].detach().cpu() - torch.from_numpy(g["flow"])).abs().max()) < 1e-4
    grads = {k: p.grad for k, p in net.named_parameters()}
    assert list(grads) == list(g["grad_keys"]) and all(v is not None for v in grads.values())
    norms = np.array([float(v.double().norm()) for v in grads.values()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("g:"):
            assert rel_err(grads[k[2:]], torch.from_numpy(g[k])) < 2e-3, k


def test_reference_training_sequence_three_steps_vs_golden(golden):
    """The literal statement sequence of train_mvsec.py:241-258 with torch's optimizer, scheduler and GradScaler."""
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b, seed = int(g["batch"]), int(g["seed"])
    model, _ = make_net(seed)
    model.change_imagesize((h, w))
    model.train()
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, weight_decay=5e-5, eps=1e-8)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, 1e-3, 20 + 100, pct_start=0.05, cycle_momentum=False,
                                                    anneal_strategy='linear')
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    losses, lrs = [], []
    for step in range(3):
        e1, e2, gt, valid = batch_of(seed + 3100 + step, seed + 3200 + step, b, h, w)
        optimizer.zero_grad()
        _, flow_list = model(e1, e2)
        loss, metrics = T.sequence_loss(flow_list, gt, valid, 0.8)
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        lrs.append(optimizer.param_groups[0]["lr"])
        scaler.step(optimizer)
        scheduler.step()
        scaler.update()
        losses.append(loss.item())
    np.testing.assert_allclose(lrs, g["step_lrs"], rtol=1e-6)
    np.testing.assert_allclose(losses, g["step_losses"], rtol=2e-4)
    final = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    np.testing.assert_allclose([float(v.double().norm()) for v in final.values()], g["final_norms"], rtol=1e-4)
    for k in g.files:
        if k.startswith("p3:"):
            assert float((final[k[3:]] - torch.from_numpy(g[k])).abs().max()) < 2e-4, k
    # the stepped nn.Parameters are what inference now uses (device-to-device weight update, HIP graph path)
    model.eval()
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(77, 1, h, w))
    with torch.no_grad():
        flow = model(e1.to(DEV), e2.to(DEV))[1][0].cpu()
        ref, _ = O.eemflow_forward(final, e1, e2)
    assert float((flow - ref).abs().max()) < 1e-4


def test_hip_sequence_loss_equals_oracle_sequence_loss():
    b, h, w = 3, 50, 70
    rng = np.random.default_rng(5)
    preds = [torch.from_numpy(rng.standard_normal((b, 2, h, w)).astype(np.float32) * 3).to(DEV).requires_grad_(True) for _ in range(3)]
    gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(6, b, h, w))
    loss, metrics = hip_train.sequence_loss(preds, gt, valid, 0.8)
    loss.backward()
    ref_preds = [p.detach().cpu().requires_grad_(True) for p in preds]
    rloss, rmetrics = T.sequence_loss(ref_preds, gt.cpu(), valid.cpu(), 0.8)
    rloss.backward()
    assert abs(float(loss) - float(rloss)) < 1e-6
    for k in ("epe", "1px", "3px", "5px"):
        assert abs(metrics[k] - rmetrics[k]) < 1e-5, k
    for p, r in zip(preds, ref_preds):
        assert float((p.grad.cpu() - r.grad).abs().max()) < 1e-9


@pytest.mark.parametrize("b,h,w,size,mesh", [(2, 100, 150, (100, 150), False), (2, 128, 128, (128, 128), True)])
def test_autograd_path_equals_fused_trainer_and_oracle(b, h, w, size, mesh):
    oh, ow = (16, 16) if mesh else (h, w)
    e1, e2, gt, valid = batch_of(51, 52, b, h, w, oh, ow)
    net, sd = make_net(53, out_mesh_size=mesh)
    net.change_imagesize(size)
    _, preds = net(e1, e2)
    loss, _ = hip_train.sequence_loss(preds, gt, valid, 0.8)
    loss.backward()
    auto = {k: p.grad.clone() for k, p in net.named_parameters()}
    net2, _ = make_net(53, out_mesh_size=mesh)
    net2.change_imagesize(size)
    tr = EEMFlowTrainer(net2, lr=0.0, wdecay=0.0, clip=0.0)
    floss, _, fflow = tr.step(e1, e2, gt, valid)
    assert abs(float(loss) - floss) < 1e-7 and torch.equal(preds[0].detach(), fflow)
    off = 0
    for k, v in auto.items():                                    # same kernels, float atomics: summation-order round-off only
        assert rel_err(v, tr.grad[off:off + v.numel()].view_as(v)) < 2e-5, k
        off += v.numel()
    rloss, _, rgrads, _ = T.loss_and_grads(sd, e1.cpu(), e2.cpu(), gt.cpu(), valid.cpu(), image_size=size,
                                           out_size=(16, 16) if mesh else None)
    assert abs(float(loss) - rloss) < 1e-5
    worst = max((rel_err(auto[k], rgrads[k]), k) for k in sd)
    assert worst[0] < 3e-3, worst


def test_two_forwards_before_backward_and_frozen_parameters():
    """A second forward reuses the context's workspace: the first graph's backward recomputes its activations.  Parameters with
    requires_grad=False get no .grad (fetch_optimizer filters on it, train_mvsec.py:180)."""
    net, sd = make_net(61)
    net.change_imagesize((96, 128))
    for p in net.rconv_1.parameters():
        p.requires_grad_(False)
    ea, eb, gt, valid = batch_of(62, 63, 2, 96, 128)
    ec, ed, gt2, valid2 = batch_of(64, 65, 2, 96, 128)
    _, pa = net(ea, eb)
    _, pc = net(ec, ed)
    la, _ = T.sequence_loss(pa, gt, valid)
    lc, _ = T.sequence_loss(pc, gt2, valid2)
    (la + 2.0 * lc).backward()
    _, _, ga, _ = T.loss_and_grads(sd, ea.cpu(), eb.cpu(), gt.cpu(), valid.cpu())
    _, _, gc, _ = T.loss_and_grads(sd, ec.cpu(), ed.cpu(), gt2.cpu(), valid2.cpu())
    for k, p in net.named_parameters():
        if k.startswith("rconv_1."):
            assert p.grad is None
        else:
            assert rel_err(p.grad, ga[k] + 2.0 * gc[k]) < 3e-3, k


def test_eval_without_grad_still_replays_the_graph_and_data_writes_need_invalidate():
    net, sd = make_net(71)
    net.change_imagesize((64, 96))
    net.eval()
    e1, e2, _, _ = batch_of(72, 73, 1, 64, 96)
    with torch.no_grad():
        f0 = net(e1, e2)[1][0].clone()
        assert not f0.requires_grad
        with torch.no_grad():
            net.out_conv.bias.data.add_(1.0)                     # bypasses the version counter
        net.invalidate_weights()
        f1 = net(e1, e2)[1][0]
    assert float((f1 - f0 - 1.0).abs().max()) < 1e-5
    # grad mode with eval(): still differentiable (the reference's module is, whatever .training says)
    _, preds = net(e1, e2)
    assert preds[0].requires_grad and torch.equal(preds[0].detach(), f1)


@pytest.mark.parametrize("val_batch,val_hw", [(1, (96, 128)), (3, (128, 192))])
def test_validation_forward_between_training_forward_and_backward(val_batch, val_hw):
    """`_, p = model(e1, e2)` under grad, then a `torch.no_grad()` forward of the same module (another batch size / image size: the
    validation step of a training loop), then `loss.backward()`: the inference forward overwrote the shared workspace, so the
    backward has to notice (newer serial) and recompute - with the TRAINING forward's shape, not the last forward's."""
    net, sd = make_net(81)
    net.change_imagesize((96, 128))
    e1, e2, gt, valid = batch_of(82, 83, 2, 96, 128)
    _, preds = net(e1, e2)
    loss, _ = T.sequence_loss(preds, gt, valid)
    with torch.no_grad():
        net.change_imagesize(val_hw)
        v1, v2, _, _ = batch_of(84, 85, val_batch, *val_hw)
        net(v1, v2)
        net(v1, v2)                                               # the second one is a graph replay
        net.change_imagesize((96, 128))
    loss.backward()
    _, _, ref, _ = T.loss_and_grads(sd, e1.cpu(), e2.cpu(), gt.cpu(), valid.cpu())
    worst = max((rel_err(p.grad, ref[k]), k) for k, p in net.named_parameters())
    assert worst[0] < 3e-3, worst
    # and the serial protocol itself: an inference forward invalidates a pending eemflow_backward
    import ctypes
    from eemflow_amd import _lib
    L, ctx = _lib.lib(), net._ctx
    flow = torch.empty(2, 2, 96, 128, device=DEV)
    serial = ctypes.c_int64()
    sp = _lib.current_stream_ptr(torch.device(DEV))
    _lib.check(L.eemflow_forward_train(ctx, e1.data_ptr(), e2.data_ptr(), 2, 96, 128, flow.data_ptr(), 96, 128, ctypes.byref(serial), sp))
    _lib.check(L.eemflow_forward(ctx, e1.data_ptr(), e2.data_ptr(), 2, 96, 128, flow.data_ptr(), 96, 128, sp))
    grad = torch.empty(sum(p.numel() for p in net.parameters()), device=DEV)
    assert L.eemflow_backward(ctx, serial.value, e1.data_ptr(), e2.data_ptr(), flow.data_ptr(), grad.data_ptr(), sp) != 0
    assert "eemflow_forward_train" in L.eemflow_last_error().decode()
