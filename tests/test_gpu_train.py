"""EEMFlow training step on the GPU (C ABI) against the training oracle (torch autograd through the functional
oracle) and the reference-generated golden.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from eemflow_amd import EEMFlow
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


def split_flat(flat, sd):
    out, off = {}, 0
    for k, v in sd.items():
        out[k] = flat[off:off + v.numel()].view_as(v)
        off += v.numel()
    return out


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_grads(net, e1, e2, gt, valid):
    tr = EEMFlowTrainer(net, lr=0.0, wdecay=0.0, clip=0.0)       # lr 0: the step leaves the weights unchanged
    loss, metrics, flow = tr.step(e1.to(DEV), e2.to(DEV), gt.to(DEV), valid.to(DEV))
    return loss, metrics, flow.cpu(), tr.grad.clone().cpu()


def test_loss_and_gradients_vs_golden(golden):
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b = int(g["batch"])
    net, sd = make_net(int(g["seed"]))
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(int(g["gt_seed"]), b, h, w))
    loss, metrics, flow, flat = run_grads(net, e1, e2, gt, valid)
    assert abs(loss - float(g["loss"])) < 1e-5 and abs(metrics["epe"] - float(g["epe"])) < 1e-4
    assert float((flow - torch.from_numpy(g["flow"])).abs().max()) < 1e-4
    grads = split_flat(flat, sd)
    norms = np.array([float(v.double().norm()) for v in grads.values()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("g:"):
            assert rel_err(grads[k[2:]], torch.from_numpy(g[k])) < 2e-3, k


@pytest.mark.parametrize("b,h,w,size", [(3, 70, 100, (70, 100)), (1, 128, 192, (128, 192)), (2, 64, 64, (100, 120))])
def test_gradients_vs_oracle_autograd(b, h, w, size):
    net, sd = make_net(23)
    net.change_imagesize(size)                                   # (the third case: padder built for another size)
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(24, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(25, b, h, w))
    loss, _, flow, flat = run_grads(net, e1, e2, gt, valid)
    rloss, _, rgrads, rflow = T.loss_and_grads(sd, e1, e2, gt, valid, image_size=size)
    assert abs(loss - rloss) < 1e-5 and float((flow - rflow).abs().max()) < 1e-4
    grads = split_flat(flat, sd)
    errs = sorted(((rel_err(grads[k], rgrads[k]), k) for k in sd), reverse=True)
    # LeakyReLU is not differentiable at 0: a pre-activation that the two fp32 summation orders put on different sides of 0 changes one
    # pixel's derivative from 1 to 0.1 and with it ONE layer's weight and bias gradient (measured: one such pixel of pconv3_1's 16x24
    # map moves 77 of that layer's 18 432 weight gradients by > 1e-3 of the largest, 3.8e-3 at worst, relative L2 1.5e-3).  So: every
    # tensor within 3e-3 in the max norm except at most one layer's pair, and those within 8e-3 with a relative L2 error < 3e-3.
    assert all(e < 3e-3 for e, _ in errs[2:]), errs[:4]
    for e, k in errs[:2]:
        l2 = float((grads[k].double().cpu() - rgrads[k].double()).norm() / (rgrads[k].double().norm() + 1e-12))
        assert e < 8e-3 and l2 < 3e-3, (e, l2, k)
    if errs[1][0] >= 3e-3:
        assert errs[0][1].rsplit(".", 1)[0] == errs[1][1].rsplit(".", 1)[0], errs[:2]


def test_decoder_winograd_streams_follow_the_optimizer_steps(monkeypatch):
    """The decoders' conv1 / conv5 run on the Winograd kernel from four samples per launch on where the 1/64 grid's rows are 16-byte
    multiples (256 x 256: 4 x 4 cells); its streams are packed on the device from the flat weights, so every optimizer step has to mark
    them stale (refresh_wino) and the next forward - training or validation - re-pack them: three steps with a real learning rate, then the
    flow of a validation forward against the same weights through the other kernel (EEM_DEC_WNC is read per call; graphs off) - stale
    streams would be off by the three steps' updates - and the two forms' weight trajectories against each other: the forms differ in
    rounding only, so the trajectories stay within a percent of the distance travelled."""
    b, h, w = 4, 256, 256
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(71, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(72, b, h, w))
    flows, weights = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("EEM_DEC_WNC", mode)
        net, sd = make_net(73)
        net.use_graph = False
        net.change_imagesize((h, w))
        tr = EEMFlowTrainer(net, lr=2e-2, wdecay=1e-4, clip=1.0)          # (OneCycle starts at lr / 25: ~8e-4 per step)
        for _ in range(3):
            tr.step(e1.to(DEV), e2.to(DEV), gt.to(DEV), valid.to(DEV))
        with torch.no_grad():
            after = net(e1.to(DEV), e2.to(DEV))[1][0].clone()
            monkeypatch.setenv("EEM_DEC_WNC", "0" if mode == "1" else "1")            # the same weights through the other kernel
            other = net(e1.to(DEV), e2.to(DEV))[1][0].clone()
            monkeypatch.setenv("EEM_DEC_WNC", mode)
        assert not torch.equal(after, other)                                       # (the switch did switch)
        assert float((after - other).abs().max()) < 2e-5, mode                      # stale streams would be off by the three steps' updates
        flows[mode] = after.cpu()
        tr.sync_parameters()
        weights[mode] = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    fresh = {k: torch.as_tensor(v).cpu() for k, v in sd.items()}
    moved = max(float((weights["1"][k] - fresh[k]).abs().max()) for k in fresh)
    drift = max(float((weights["1"][k] - weights["0"][k]).abs().max()) for k in fresh)
    assert moved > 1e-3 and drift < 1e-2 * moved                                    # both forms train the same model
    assert float((flows["1"] - flows["0"]).abs().max()) < 5e-2 * max(1.0, float(flows["0"].abs().max()))


def test_weight_gradients_on_the_side_stream_equal_the_single_stream_pass(monkeypatch):
    """Weight / bias gradients run on a context-owned side stream behind events (they are leaves of the data-gradient chain);
    EEM_NO_WGRAD_STREAM=1 keeps them on the caller's stream.  Same gradients (summation order of the split-K atomics aside), repeated
    steps included: the next forward must not overwrite an activation a pending weight-gradient launch still reads."""
    b, h, w = 4, 260, 346
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(61, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(62, b, h, w))
    out = {}
    for mode in ("side", "single"):
        monkeypatch.setenv("EEM_NO_WGRAD_STREAM", "1" if mode == "single" else "0")
        net, sd = make_net(63)
        net.change_imagesize((h, w))
        tr = EEMFlowTrainer(net, lr=0.0, wdecay=0.0, clip=0.0)
        grads = []
        with torch.cuda.stream(torch.cuda.Stream()):                  # a caller stream other than the default one
            for _ in range(3):
                loss, _, flow = tr.step(e1.to(DEV), e2.to(DEV), gt.to(DEV), valid.to(DEV))
                grads.append(tr.grad.clone())
            torch.cuda.current_stream().synchronize()
        out[mode] = (loss, flow.cpu(), [g.cpu() for g in grads])
    assert abs(out["side"][0] - out["single"][0]) < 1e-12 * abs(out["single"][0])      # f64 atomics of the loss kernel: order only
    assert torch.equal(out["side"][1], out["single"][1])
    ref = out["single"][2][0]
    for g in out["side"][2] + out["single"][2][1:]:
        assert float((g - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("b,h,w", [(2, 260, 346), (1, 720, 1280), (3, 200, 300)])
def test_dedicated_backward_kernels_equal_generic_ones(monkeypatch, b, h, w):
    """wgrad_enc / dgrad_s2 / wgrad_small against the generic conv and weight-gradient kernels of the same library
    (EEM_NO_* are read at launch): every one of the 66 gradient tensors to summation-order round-off."""
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(41, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(42, b, h, w))

    def run():
        net, sd = make_net(43)
        net.change_imagesize((h, w))
        loss, _, flow, flat = run_grads(net, e1, e2, gt, valid)
        return loss, flow, split_flat(flat, sd)
    loss_f, flow_f, fast = run()
    for k in ("EEM_NO_WGRAD_ENC", "EEM_NO_DGRAD_S2", "EEM_NO_WGRAD_SMALL", "EEM_NO_WGRAD_TAIL"):
        monkeypatch.setenv(k, "1")
    loss_g, flow_g, gen = run()
    assert abs(loss_f - loss_g) < 1e-9 and torch.equal(flow_f, flow_g)   # same forward code; the loss sums by f64 atomics
    worst = max((rel_err(fast[k], gen[k]), k) for k in fast)
    assert worst[0] < 2e-5, worst


@pytest.mark.parametrize("b,h,w,groups", [(32, 260, 346, 5), (8, 720, 1280, 5), (3, 200, 300, 5), (5, 260, 346, 2), (2, 128, 192, 1)])
def test_tail_weight_gradients_in_one_launch_equal_the_per_layer_launches(monkeypatch, b, h, w, groups):
    """wgrad_tail.hip (round 6): the weight and bias gradients of every 3x3 conv of the 1/64-grid tail - three decoders x seven layers with
    their groups, the three rconv_k (EEMFlow.py:37-69,96-102 under train_mvsec.py:253) - as ONE launch behind the tail's backward chain,
    against the per-layer wgrad_small / bias launches (EEM_NO_WGRAD_TAIL=1, read per call): all 66 gradient tensors to summation-order
    round-off.  BASELINE configs[2] and [3] at their stated batch, a ragged batch (K parts of unequal size), other group counts."""
    from eemflow_amd import EEMFlow
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(44, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(45, b, h, w))

    def run():
        sd = O.to_torch_sd(seeded_state_dict(46, groups=groups))
        net = EEMFlow("", groups=groups, n_first_channels=5)
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        net.change_imagesize((h, w))
        loss, _, flow, flat = run_grads(net, e1, e2, gt, valid)
        return loss, split_flat(flat, sd)
    monkeypatch.delenv("EEM_NO_WGRAD_TAIL", raising=False)
    loss_f, fast = run()
    monkeypatch.setenv("EEM_NO_WGRAD_TAIL", "1")
    loss_g, gen = run()
    assert abs(loss_f - loss_g) < 1e-9
    assert not all(torch.equal(fast[k], gen[k]) for k in fast if k.startswith("decoder_"))       # (the switch did switch)
    worst = max((rel_err(fast[k], gen[k]), k) for k in fast)
    assert worst[0] < 2e-5, worst


def test_three_optimizer_steps_vs_golden(golden):
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b, seed = int(g["batch"]), int(g["seed"])
    net, sd = make_net(seed)
    net.change_imagesize((h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    losses, lrs = [], []
    for step in range(3):
        e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(seed + 3100 + step, b, h, w))
        gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(seed + 3200 + step, b, h, w))
        loss, m, _ = tr.step(e1, e2, gt, valid)
        losses.append(loss)
        lrs.append(m["lr"])
    np.testing.assert_allclose(lrs, g["step_lrs"], rtol=1e-6)
    np.testing.assert_allclose(losses, g["step_losses"], rtol=2e-4)
    tr.sync_parameters()
    final = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    np.testing.assert_allclose([float(v.double().norm()) for v in final.values()], g["final_norms"], rtol=1e-4)
    for k in g.files:
        if k.startswith("p3:"):
            assert float((final[k[3:]] - torch.from_numpy(g[k])).abs().max()) < 2e-4, k
    # the updated weights are the ones inference now uses
    net.eval()
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(77, 1, h, w))
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0].cpu()
        ref, _ = O.eemflow_forward({k: v for k, v in final.items()}, e1, e2)
    assert float((flow - ref).abs().max()) < 1e-4


def test_non_finite_gradient_skips_the_step_like_gradscaler():
    """train_mvsec.py:237,257: `scaler.step(optimizer)` skips the step when a gradient holds an inf / NaN - weights, moments and the
    optimizer's own step count (the bias corrections) stay put.  eemflow_optimizer_step does the same: two poisoned calls between real
    steps change nothing, and the three real steps equal torch's AdamW + clip_grad_norm_ fed the same three gradients."""
    from eemflow_amd import _lib
    h, w, b = 128, 192, 2
    net, sd = make_net(61)
    net.change_imagesize((h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    ref = torch.nn.Parameter(torch.cat([v.reshape(-1) for v in sd.values()]).clone().to(DEV))
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=5e-5, eps=1e-8)
    L, sp = _lib.lib(), _lib.current_stream_ptr(torch.device(DEV))

    def weights():
        tr.sync_parameters()
        return torch.cat([v.reshape(-1).float() for v in net.state_dict().values()]).to(DEV)

    assert tr.skipped_steps() == 0
    for step in range(3):
        e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(62 + step, b, h, w))
        gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(70 + step, b, h, w))
        _, m, _ = tr.step(e1, e2, gt, valid)
        g = tr.grad.clone()
        assert bool(torch.isfinite(g).all())
        for q in opt.param_groups:
            q["lr"] = m["lr"]
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        if step == 0:                                     # two poisoned steps straight through the C entry point
            w0 = weights()
            for poison in (float("inf"), float("nan")):
                bad = g.clone()
                bad[12345] = poison
                _lib.check(L.eemflow_optimizer_step(net._ctx, bad.data_ptr(), 1e-3, 5e-5, 1e-8, 1.0, sp))
            assert tr.skipped_steps() == 2
            assert torch.equal(weights(), w0)
    assert tr.skipped_steps() == 2
    d = (weights() - ref.detach()).abs()
    assert float(d.max()) < 2e-6, float(d.max())          # bias corrections of steps 1, 2, 3 - not 1, 4, 5


def test_reload_after_a_skipped_step_restarts_the_skip_count():
    """eemflow_load_weights restarts the optimisation (step count, moments): the device-side count of skipped steps must restart with
    it, or the first steps after the reload compute their bias corrections from `step - skipped <= 0` (inf / NaN / sign-flipped
    updates).  A skipped step, a reload of the same weights, then one real step must equal torch's first AdamW step."""
    from eemflow_amd import _lib
    h, w, b = 128, 192, 1
    net, sd = make_net(63)
    net.change_imagesize((h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    L, sp = _lib.lib(), _lib.current_stream_ptr(torch.device(DEV))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(64, b, h, w))
    gt, valid = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(65, b, h, w))
    tr.step(e1, e2, gt, valid)
    bad = tr.grad.clone()
    bad[7] = float("nan")
    for _ in range(3):
        _lib.check(L.eemflow_optimizer_step(net._ctx, bad.data_ptr(), 1e-3, 5e-5, 1e-8, 1.0, sp))
    assert tr.skipped_steps() == 3
    host0 = torch.cat([v.reshape(-1) for v in sd.values()]).clone().contiguous()          # eemflow_load_weights takes a HOST vector
    _lib.check(L.eemflow_load_weights(net._ctx, host0.data_ptr(), host0.numel(), 5, 5))
    flat0 = host0.to(DEV)
    assert tr.skipped_steps() == 0
    g = torch.randn_like(flat0) * 1e-3
    _lib.check(L.eemflow_optimizer_step(net._ctx, g.data_ptr(), 1e-3, 5e-5, 1e-8, 1.0, sp))
    got = torch.empty_like(flat0)
    _lib.check(L.eemflow_get_weights(net._ctx, got.data_ptr(), got.numel(), sp))
    torch.cuda.synchronize()
    ref = torch.nn.Parameter(flat0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=5e-5, eps=1e-8)
    ref.grad = g.clone()
    torch.nn.utils.clip_grad_norm_([ref], 1.0)
    opt.step()
    assert bool(torch.isfinite(got).all())
    assert float((got - ref.detach()).abs().max()) < 2e-6


@pytest.mark.parametrize("groups", [1, 2, 4])
def test_decoder_groups_other_than_five(groups):
    """EEMFlow(config, groups=g) (EEMFlow.py:72, Decoder(69, groups) :37-47): every divisor of 100 up to the reference's default 5 -
    inference against the oracle, and loss + all gradient tensors of the fused training step against torch autograd through the oracle."""
    b, h, w = 2, 128, 192
    sd = seeded_state_dict(71, groups=groups)
    net = EEMFlow("", groups=groups, n_first_channels=5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(DEV)
    tsd = O.to_torch_sd(sd)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(72, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(73, b, h, w))
    net.eval()
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0].cpu()
        ref, _ = O.eemflow_forward(tsd, e1, e2, groups=groups)
    assert float((flow - ref).abs().max()) < 1e-4
    net.train()
    loss, _, tflow, flat = run_grads(net, e1, e2, gt, valid)
    rloss, _, rgrads, rflow = T.loss_and_grads(tsd, e1, e2, gt, valid, groups=groups)
    assert abs(loss - rloss) < 1e-5 and float((tflow - rflow).abs().max()) < 1e-4
    grads = split_flat(flat, tsd)
    errs = sorted(((rel_err(grads[k], rgrads[k]), k) for k in tsd), reverse=True)
    assert all(e < 3e-3 for e, _ in errs[2:]), errs[:4]            # (one LeakyReLU unit at 0 may move one layer's pair: see above)
    assert errs[0][0] < 8e-3, errs[:2]
    # the reference's own training statements through autograd (EEMFlow.forward as a torch.autograd.Function) give the same gradients
    net.zero_grad()
    from eemflow_amd.train import sequence_loss
    out = net(e1.to(DEV), e2.to(DEV))[1]
    l2, _ = sequence_loss(out, gt.to(DEV), valid.to(DEV), 0.8)
    l2.backward()
    named = dict(net.named_parameters())
    worst = max(rel_err(named[k].grad, grads[k]) for k in tsd)
    assert worst < 1e-5, worst


@pytest.mark.parametrize("cin", [3, 15])
def test_n_first_channels_other_than_five(cin):
    """EEMFlow(config, n_first_channels=c) (EEMFlow.py:72,75; EEMFlow+'s data configuration feeds 15 bins): pconv1_1 on the generic
    convolution behind a replicate-pad launch - inference against the oracle (an image size that pads on every side but the top), and the
    fused training step's loss and gradients against torch autograd through the oracle."""
    b, h, w = 2, 100, 150
    sd = seeded_state_dict(81, n_first_channels=cin)
    net = EEMFlow("", groups=5, n_first_channels=cin)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(DEV)
    tsd = O.to_torch_sd(sd)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(82, b, h, w, bins=cin))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(83, b, h, w))
    net.eval()
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0].cpu()
        flow2 = net(e1.to(DEV), e2.to(DEV))[1][0].cpu()
        ref, st = O.eemflow_forward(tsd, e1, e2, keep=True)
    assert torch.equal(flow, flow2)
    assert float((net.stage("f11")[:b].cpu() - st["f11"]).abs().max()) < 1e-4
    assert float((flow - ref).abs().max()) < 1e-4
    net.train()
    loss, _, tflow, flat = run_grads(net, e1, e2, gt, valid)
    rloss, _, rgrads, rflow = T.loss_and_grads(tsd, e1, e2, gt, valid)
    assert abs(loss - rloss) < 1e-5 and float((tflow - rflow).abs().max()) < 1e-4
    grads = split_flat(flat, tsd)
    errs = sorted(((rel_err(grads[k], rgrads[k]), k) for k in tsd), reverse=True)
    assert all(e < 3e-3 for e, _ in errs[2:]), errs[:4]
    assert errs[0][0] < 8e-3, errs[:2]
    assert rel_err(grads["pconv1_1.0.weight"], rgrads["pconv1_1.0.weight"]) < 3e-3


def test_out_mesh_size_training():
    net, sd = make_net(29, out_mesh_size=True)
    net.change_imagesize((128, 128))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(30, 2, 128, 128))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(31, 2, 16, 16))
    loss, _, flow, flat = run_grads(net, e1, e2, gt, valid)
    rloss, _, rgrads, rflow = T.loss_and_grads(sd, e1, e2, gt, valid, out_size=(16, 16))
    assert flow.shape == (2, 2, 16, 16) and abs(loss - rloss) < 1e-5
    grads = split_flat(flat, sd)
    worst = max((rel_err(grads[k], rgrads[k]), k) for k in sd)
    assert worst[0] < 3e-3, worst


def test_loss_statistics_behind_an_event_equal_the_synchronous_read():
    """eemflow_train_stats_async / _wait (what EEMFlowTrainer.step uses: the five statistics are read after the optimizer step is
    enqueued) against eemflow_forward_backward's own synchronous stats_out on the same forward; _wait without an outstanding _async and
    _async before any training forward are argument errors."""
    import ctypes

    from eemflow_amd import _lib
    h, w, b = 128, 160, 3
    net, _ = make_net(7)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(8, b, h, w))
    gt, va = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(9, b, h, w))
    L, ctx, s = _lib.lib(), net._context(torch.device(DEV)), _lib.current_stream_ptr(torch.device(DEV))
    n = sum(p.numel() for p in net.parameters())
    grad, flow = torch.empty(n, device=DEV), torch.empty(b, 2, h, w, device=DEV)
    sync, later = (ctypes.c_double * 5)(), (ctypes.c_double * 5)()
    with pytest.raises(_lib.EEMFlowHipError, match="no eemflow_forward_backward"):
        _lib.check(L.eemflow_train_stats_async(ctx, s))
    _lib.check(L.eemflow_forward_backward(ctx, e1.data_ptr(), e2.data_ptr(), gt.data_ptr(), va.data_ptr(), b, h, w, h, w, 1.0,
                                          flow.data_ptr(), grad.data_ptr(), ctypes.byref(sync), s))
    with pytest.raises(_lib.EEMFlowHipError, match="outstanding"):
        _lib.check(L.eemflow_train_stats_wait(ctx, later))
    _lib.check(L.eemflow_forward_backward(ctx, e1.data_ptr(), e2.data_ptr(), gt.data_ptr(), va.data_ptr(), b, h, w, h, w, 1.0,
                                          flow.data_ptr(), grad.data_ptr(), None, s))
    _lib.check(L.eemflow_train_stats_async(ctx, s))
    _lib.check(L.eemflow_optimizer_step(ctx, grad.data_ptr(), 0.0, 0.0, 1e-8, 1.0, s))
    _lib.check(L.eemflow_train_stats_wait(ctx, later))
    assert list(sync) == list(later) and sync[0] > 0 and sync[2] > 0
