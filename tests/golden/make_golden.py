#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, CPU, python -B); the GPU
box never sees the reference, only the .npz files written here.  Inputs and
weights are drawn from numpy PCG64 seeds (eemflow_amd/weights.py), so fixtures
hold seeds + expected outputs, not inputs.

What is imported from the reference:
  model/EEMFlow/EEMFlow.py        EEMFlow, with net.index patched to the 53-tap list
  utils/image_utils.py            InputPadder
  loader/loader_utils.py          EventSequence, EventSequenceToVoxelGrid_Pytorch
Stubs written here (NOT reference code) for packages the image lacks:
  utils_luo.tools                 unimportable (cv2/imageio/png, torch whitelist) - only
                                  `tensor_tools.check_tensor` is referenced, in demo()
  spatial_correlation_sampler     third-party, pinned ==0.4.0 in requirements.txt:131, not
                                  vendored: restated from its documented semantics
                                  out[b,ph,pw,h,w] = sum_c x[b,c,h,w]*y[b,c,h+ph-4,w+pw-4]
                                  => the local-correlation goldens are "parity unpinned"
                                  against the real extension.
  cv2 / torchvision / h5py        imported at module scope by loader_utils, unused by the voxelizer

Usage:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from eemflow_amd.weights import (CORR_TAPS_53, seeded_state_dict,  # noqa: E402
                                 synthetic_voxel_pair)


# --------------------------------------------------------------------------- stubs
def install_stubs():
    ul = types.ModuleType("utils_luo")
    ul.__path__ = []
    ult = types.ModuleType("utils_luo.tools")

    class tools:  # noqa: N801
        pass

    class tensor_tools:  # noqa: N801
        @classmethod
        def check_tensor(cls, data, name, **kw):
            print(name, tuple(data.shape))

    ult.tools, ult.tensor_tools = tools, tensor_tools
    ul.tools = ult
    sys.modules["utils_luo"], sys.modules["utils_luo.tools"] = ul, ult

    scs = types.ModuleType("spatial_correlation_sampler")

    class SpatialCorrelationSampler(nn.Module):
        def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
            super().__init__()
            assert (kernel_size, stride, padding, dilation, dilation_patch) == (1, 1, 0, 1, 1)
            self.patch = patch_size

        def forward(self, x, y):
            b, c, h, w = x.shape
            r = self.patch // 2
            yp = F.pad(y, (r, r, r, r))
            out = [(x * yp[:, :, i:i + h, j:j + w]).sum(1) for i in range(self.patch) for j in range(self.patch)]
            return torch.stack(out, 1).view(b, self.patch, self.patch, h, w)

    scs.SpatialCorrelationSampler = SpatialCorrelationSampler
    sys.modules["spatial_correlation_sampler"] = scs

    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class ColorJitter:
        def __init__(self, *a, **k):
            pass

    tvt.ColorJitter = ColorJitter
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    sys.modules["h5py"] = types.ModuleType("h5py")


def load_reference():
    install_stubs()
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("EEMFlow_ref", f"{REF}/model/EEMFlow/EEMFlow.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.path.insert(0, f"{REF}/loader")
    import loader.loader_utils as lu
    from utils.image_utils import InputPadder
    return mod, lu, InputPadder


def ref_eemflow(mod, sd_np, n_first_channels=5):
    net = mod.EEMFlow(config="", groups=5, n_first_channels=n_first_channels).eval()
    net.index = torch.tensor(CORR_TAPS_53)          # SURVEY finding 1
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- fixtures
def gen_pad(InputPadder):
    sizes = [(260, 346), (720, 1280), (480, 640), (512, 960), (128, 192), (64, 64), (65, 127)]
    rows = []
    for (h, w) in sizes:
        for rate in (64, 32):
            for mode in ("chairs", "sintel"):
                p = InputPadder((h, w), mode=mode, eval_pad_rate=rate)
                rows.append([h, w, rate, 0 if mode == "chairs" else 1] + list(p._pad))
    x = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).view(2, 3, 5, 7)
    p = InputPadder((5, 7), mode="chairs", eval_pad_rate=4)
    xp = p.pad(x)[0]
    save("pad.npz", table=np.array(rows, dtype=np.int64), x=x.numpy(), x_padded=xp.numpy(),
         x_pad=np.array(p._pad), x_unpadded=p.unpad(xp).numpy())


def gen_eemflow(mod, tag, seed, batch, h, w, keep_stages):
    sd = seeded_state_dict(seed)
    net = ref_eemflow(mod, sd)
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 1000, batch, h, w))
    net.change_imagesize((h, w))
    with torch.no_grad():
        (_, _), preds = net(e1, e2)
        flow = preds[0]
        # Stage tensors, recomputed with the reference's own submodules in the reference's
        # own order (EEMFlow.py:134-180) and cross-checked against the forward above.
        p1, p2 = net.image_padder.pad(e1, e2)
        f11 = net.pconv1_2(net.pconv1_1(p1)); f21 = net.pconv1_2(net.pconv1_1(p2))
        f12 = net.pconv2_3(net.pconv2_2(net.pconv2_1(f11))); f22 = net.pconv2_3(net.pconv2_2(net.pconv2_1(f21)))
        f13 = net.pconv3_3(net.pconv3_2(net.pconv3_1(f12))); f23 = net.pconv3_3(net.pconv3_2(net.pconv3_1(f22)))
        pooled = {}
        out = {}
        flows = []
        for k, (fa, fb, ps) in enumerate(((f11, f21, 32), (f12, f22, 16), (f13, f23, 8)), start=1):
            pa = F.avg_pool2d(fa, kernel_size=(ps, ps), stride=(ps, ps))
            pb = F.avg_pool2d(fb, kernel_size=(ps, ps), stride=(ps, ps))
            cv = torch.index_select(net.corr(pa, pb), dim=1, index=net.index.long())
            r = getattr(net, f"rconv_{k}")(pa)
            fl = getattr(net, f"decoder_{k}")(torch.cat([cv, r], 1))
            pooled[f"pool1_{k}"], pooled[f"pool2_{k}"] = pa.numpy(), pb.numpy()
            out[f"cv_{k}"], out[f"r_{k}"], out[f"flow_{k}"] = cv.numpy(), r.numpy(), fl.numpy()
            flows.append(fl)
        coarse = net.out_conv(torch.cat(flows, 1))
        flow2 = net.upsample_flow(coarse, (h, w))
        assert torch.equal(flow, flow2), "stage recomputation diverged from reference forward"
    arrays = dict(seed=np.int64(seed), input_seed=np.int64(seed + 1000), batch=np.int64(batch),
                  hw=np.array([h, w]), pad=np.array(net.image_padder._pad),
                  coarse=coarse.numpy(), flow=flow.numpy(), **pooled, **out)
    if keep_stages:
        arrays.update(f11=f11.numpy(), f12=f12.numpy(), f13=f13.numpy(), f23=f23.numpy())
    save(f"eemflow_fwd_{tag}.npz", **arrays)


def gen_layout(mod):
    """Key names, order and shapes of the reference module's state_dict (EEMFlow.py:72-112)."""
    net = mod.EEMFlow(config="", groups=5, n_first_channels=5)
    sd = net.state_dict()
    shapes = np.full((len(sd), 4), -1, dtype=np.int64)
    for i, v in enumerate(sd.values()):
        shapes[i, :v.dim()] = list(v.shape)
    save("state_dict_layout.npz", keys=np.array(list(sd.keys())), shapes=shapes,
         nparams=np.int64(sum(v.numel() for v in sd.values())))


def gen_corr(mod):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 8, 7, 9), dtype=np.float32)
    y = rng.standard_normal((2, 8, 7, 9), dtype=np.float32)
    corr = mod.Correlation(4)
    with torch.no_grad():
        cv81 = corr(torch.from_numpy(x), torch.from_numpy(y))
        cv53 = torch.index_select(cv81, 1, torch.tensor(CORR_TAPS_53))
    save("local_corr.npz", x=x, y=y, cv81=cv81.numpy(), cv53=cv53.numpy())


def gen_decoder(mod):
    """Decoder alone (grouped convs + channel shuffle, EEMFlow.py:37-69)."""
    sd = seeded_state_dict(3)
    net = ref_eemflow(mod, sd)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 69, 5, 6), dtype=np.float32)
    with torch.no_grad():
        xt = torch.from_numpy(x)
        d = net.decoder_2
        c1 = d.conv1(xt)
        c2 = d.channel_shuffle(d.conv2(c1), 5)
        y = d(xt)
        shuf = d.channel_shuffle(torch.arange(100, dtype=torch.float32).view(1, 100, 1, 1), 5).flatten()
    save("decoder.npz", seed=np.int64(3), x=x, conv1=c1.numpy(), conv2_shuffled=c2.numpy(), y=y.numpy(),
         shuffle_perm=shuf.numpy().astype(np.int64))


def make_events(rng, n, h, w, t_span, pol01=False, ties=False):
    t = np.sort(rng.uniform(0.0, t_span, size=n))
    if ties and n > 8:
        t[n // 4: n // 4 + 5] = t[n // 4]          # timestamp ties
        t[-3:] = t[-1]                              # several events at t_last
    x = rng.integers(0, w, size=n).astype(np.float64)
    y = rng.integers(0, h, size=n).astype(np.float64)
    p = rng.integers(0, 2, size=n).astype(np.float64)
    if not pol01:
        p = p * 2 - 1
    return np.stack([t, x, y, p], axis=1)


def gen_voxel(lu):
    """Voxel grids + the int64 flat indices of both index_add_ calls (loader_utils.py:447-537)."""
    cases = {}
    rng = np.random.default_rng(21)
    h, w, bins = 48, 64, 5

    def run(name, ev, hh=h, ww=w, normalize=True, nb=bins):
        seq = lu.EventSequence(None, {"height": hh, "width": ww}, features=ev.copy(),
                               timestamp_multiplier=1e6, convert_to_relative=True)
        vox = lu.EventSequenceToVoxelGrid_Pytorch(num_bins=nb, normalize=normalize, gpu=False, forkserver=False)
        grid = vox(seq)
        cases[f"{name}_events"] = ev
        cases[f"{name}_hwb"] = np.array([hh, ww, nb, int(normalize)])
        cases[f"{name}_grid"] = grid.numpy()
        # integer indices, recomputed with the reference's exact expressions (:488-523) on the
        # sequence the reference built (sorted, scaled by 1e6, made relative)
        f = torch.from_numpy(seq.features.astype("float"))
        dT = f[-1, 0] - f[0, 0]
        dT = 1.0 if dT == 0 else dT
        ts = (nb - 1) * (f[:, 0] - f[0, 0]) / dT
        xs, ys = f[:, 1].long(), f[:, 2].long()
        tis = torch.floor(ts)
        tl = tis.long()
        vl = (tis < nb) & (tis >= 0)
        vr = ((tis + 1) < nb) & (tis >= 0)
        cases[f"{name}_idx_left"] = (xs[vl] + ys[vl] * ww + tl[vl] * ww * hh).numpy()
        cases[f"{name}_idx_right"] = (xs[vr] + ys[vr] * ww + (tl[vr] + 1) * ww * hh).numpy()

    run("n20k", make_events(rng, 20000, h, w, 0.05, ties=True))
    run("n20k_pol01", make_events(rng, 20000, h, w, 0.05, pol01=True, ties=True))
    run("n1", make_events(rng, 1, h, w, 0.05))
    run("n2_dt0", np.array([[0.01, 3, 4, 1.0], [0.01, 3, 4, -1.0]]))            # deltaT == 0, cancels to zero
    run("n3_dt0", np.array([[0.02, 5, 6, 1.0], [0.02, 5, 6, 1.0], [0.02, 7, 1, -1.0]]))
    run("n500_raw", make_events(rng, 500, h, w, 0.05), normalize=False)
    run("n300_bins3", make_events(rng, 300, 20, 24, 0.01), hh=20, ww=24, nb=3)
    ev = make_events(rng, 400, h, w, 0.05)
    ev = ev[rng.permutation(len(ev))]                                               # unsorted input
    run("n400_unsorted", ev)
    run("n64_const", np.stack([np.linspace(0, 0.03, 64), np.arange(64) % w, np.arange(64) % h,
                               np.ones(64)], 1))                                    # all +1
    save("voxel.npz", **cases)


def ref_eraft(seed):
    """Reference ERAFT loaded (strict) with the product's seeded weights (eemflow_amd/eraft_weights.py)."""
    from model.eraft import ERAFT
    from eemflow_amd.eraft import ERAFT as Mirror
    from eemflow_amd.eraft_weights import seeded_from_shapes
    shapes = {k: tuple(v.shape) for k, v in Mirror("", 5).state_dict().items()}
    sd = seeded_from_shapes(shapes, seed)
    net = ERAFT(config="", n_first_channels=5).eval()
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net


def gen_eraft_layout():
    from model.eraft import ERAFT
    sd = ERAFT(config="", n_first_channels=5).state_dict()
    shapes = np.full((len(sd), 4), -1, dtype=np.int64)
    for i, v in enumerate(sd.values()):
        shapes[i, :v.dim()] = list(v.shape)
    save("eraft_layout.npz", keys=np.array(list(sd.keys())), shapes=shapes,
         nparams=np.int64(sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)))


def gen_eraft(tag, seed, batch, h, w, iters, keep):
    net = ref_eraft(seed)
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 2000, batch, h, w))
    net.change_imagesize((h, w))
    arrays = dict(seed=np.int64(seed), input_seed=np.int64(seed + 2000), batch=np.int64(batch), hw=np.array([h, w]),
                  iters=np.int64(iters))
    with torch.no_grad():
        (_, _), preds = net(e1, e2, iters=iters)
        arrays["pad"] = np.array(net.image_padder._pad)
        arrays["preds"] = torch.stack(preds).numpy()
        if keep:
            from model.corr import CorrBlock
            from model.model_utils import coords_grid
            im1, im2 = net.image_padder.pad(e1, e2)
            fmap1, fmap2 = net.fnet([im1.contiguous(), im2.contiguous()])
            cnet = net.cnet(im1.contiguous())
            n0, inp = torch.split(cnet, [128, 128], dim=1)
            n0, inp = torch.tanh(n0), torch.relu(inp)
            cb = CorrBlock(fmap1.float(), fmap2.float(), radius=4)
            n, _, hh, ww = fmap1.shape
            c0 = coords_grid(n, hh, ww)
            corr0 = cb(c0)
            net1, mask1, delta1 = net.update_block(n0, inp, corr0, c0 - c0)
            arrays.update(fmap1=fmap1.numpy(), fmap2=fmap2.numpy(), net0=n0.numpy(), inp=inp.numpy(), corr0=corr0.numpy(),
                          net1=net1.numpy(), mask1=mask1.numpy(), delta1=delta1.numpy(),
                          pyr1=cb.corr_pyramid[1].numpy(), pyr3=cb.corr_pyramid[3].numpy())
    save(f"eraft_fwd_{tag}.npz", **arrays)


def gen_eraft_lookup():
    """CorrBlock on small random maps with fractional, negative and out-of-range coordinates (model/corr.py)."""
    from model.corr import CorrBlock
    rng = np.random.default_rng(31)
    f1 = rng.standard_normal((2, 16, 17, 19), dtype=np.float32)
    f2 = rng.standard_normal((2, 16, 17, 19), dtype=np.float32)
    coords = rng.uniform(-6, 25, size=(2, 2, 17, 19)).astype(np.float32)
    coords[0, :, 0, 0] = [3.0, 4.0]                        # exactly on a grid point
    coords[0, :, 0, 1] = [18.0, 16.0]                      # exactly on the last column / row
    with torch.no_grad():
        cb = CorrBlock(torch.from_numpy(f1), torch.from_numpy(f2), radius=4)
        out = cb(torch.from_numpy(coords))
    save("eraft_lookup.npz", f1=f1, f2=f2, coords=coords, out=out.numpy(),
         **{f"pyr{i}": p.numpy() for i, p in enumerate(cb.corr_pyramid)})


def gen_eraft_upsample():
    from model.eraft import ERAFT
    rng = np.random.default_rng(41)
    flow = rng.standard_normal((2, 2, 5, 7), dtype=np.float32) * 3
    mask = rng.standard_normal((2, 576, 5, 7), dtype=np.float32)
    net = ERAFT(config="", n_first_channels=5)
    with torch.no_grad():
        up = net.upsample_flow(torch.from_numpy(flow), torch.from_numpy(mask))
    save("eraft_upsample.npz", flow=flow, mask=mask, up=up.numpy())


def synthetic_gt(seed, batch, h, w):
    """Smooth ground-truth flow + validity mask with holes and a few |gt| > 400 pixels (numpy PCG64)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    gt = np.stack([3 * np.sin(2 * np.pi * xx / w + 0.3), 2 * np.cos(2 * np.pi * yy / h)])[None].repeat(batch, 0)
    gt = (gt + rng.normal(0, 0.5, gt.shape)).astype(np.float32)
    gt[:, :, 0, :3] = 500.0                                  # excluded by mag < MAX_FLOW
    valid = (rng.random((batch, h, w)) < 0.8).astype(np.float32)
    return gt, valid


def gen_train(mod):
    """Reference module in train mode + the reference's own sequence_loss (train_mvsec.py:201-227, exec'd from
    source) + torch AdamW / OneCycleLR / clip_grad_norm_ as train_mvsec.py:178-183,241-258 order them."""
    import textwrap
    src = open(f"{REF}/train_mvsec.py").read().splitlines()[200:227]
    ns = {"torch": torch, "MAX_FLOW": 400}
    exec(textwrap.dedent("\n".join(src)), ns)
    ref_loss = ns["sequence_loss"]
    seed, batch, h, w = 9, 2, 64, 96
    sd = seeded_state_dict(seed)
    net = ref_eemflow(mod, sd).train()
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 3000, batch, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(seed + 3001, batch, h, w))
    (_, _), preds = net(e1, e2)
    loss, metrics = ref_loss(None, preds, gt, valid, 0.8)
    loss.backward()
    grads = {k: p.grad.detach().numpy() for k, p in net.named_parameters()}
    arrays = dict(seed=np.int64(seed), input_seed=np.int64(seed + 3000), gt_seed=np.int64(seed + 3001), batch=np.int64(batch),
                  hw=np.array([h, w]), loss=np.float64(loss.item()), epe=np.float64(metrics["epe"]),
                  px1=np.float64(metrics["1px"]), flow=preds[0].detach().numpy(),
                  grad_keys=np.array(list(grads.keys())),
                  grad_norms=np.array([np.sqrt((g.astype(np.float64) ** 2).sum()) for g in grads.values()]),
                  grad_sums=np.array([g.astype(np.float64).sum() for g in grads.values()]))
    for k in ("pconv1_1.0.weight", "pconv1_1.0.bias", "pconv2_2.0.weight", "pconv3_3.0.bias", "rconv_2.0.weight",
              "decoder_1.conv1.0.weight", "decoder_2.conv3.0.weight", "decoder_3.conv7.weight", "decoder_3.conv7.bias",
              "out_conv.weight", "out_conv.bias"):
        arrays["g:" + k] = grads[k]
    # 3 optimisation steps on 3 different batches (lr 1e-3 so the trajectory is visible in fp32)
    net = ref_eemflow(mod, sd).train()
    net.change_imagesize((h, w))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=5e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-3, 20 + 100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="linear")
    losses, lrs = [], []
    for step in range(3):
        e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 3100 + step, batch, h, w))
        gt, valid = (torch.from_numpy(a) for a in synthetic_gt(seed + 3200 + step, batch, h, w))
        opt.zero_grad()
        (_, _), preds = net(e1, e2)
        loss, _ = ref_loss(None, preds, gt, valid, 0.8)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        losses.append(loss.item())
    final = net.state_dict()
    arrays.update(step_losses=np.array(losses), step_lrs=np.array(lrs),
                  final_sums=np.array([v.double().sum().item() for v in final.values()]),
                  final_norms=np.array([v.double().norm().item() for v in final.values()]))
    for k in ("pconv1_1.0.weight", "decoder_2.conv3.0.weight", "out_conv.weight", "out_conv.bias"):
        arrays["p3:" + k] = final[k].numpy()
    save("train_step.npz", **arrays)


def load_reference_plus():
    """model/EEMFlow/EEMFlow+.py with utils_luo.tools.tensor_tools.torch_warp restated from
    utils_luo/tools.py:2262-2306 (that file itself cannot be imported - SURVEY.md Appendix A)."""
    tt = sys.modules["utils_luo.tools"].tensor_tools

    def torch_warp(cls, x, flo):
        b, c, h, w = x.size()
        xx = torch.arange(0, w).view(1, -1).repeat(h, 1).view(1, 1, h, w).repeat(b, 1, 1, 1)
        yy = torch.arange(0, h).view(-1, 1).repeat(1, w).view(1, 1, h, w).repeat(b, 1, 1, 1)
        vgrid = torch.cat((xx, yy), 1).float() + flo
        vgrid[:, 0, :, :] = 2.0 * vgrid[:, 0, :, :] / max(w - 1, 1) - 1.0
        vgrid[:, 1, :, :] = 2.0 * vgrid[:, 1, :, :] / max(h - 1, 1) - 1.0
        return F.grid_sample(x, vgrid.permute(0, 2, 3, 1), padding_mode="zeros")

    tt.torch_warp = classmethod(torch_warp)
    spec = importlib.util.spec_from_file_location("EEMFlowP_ref", f"{REF}/model/EEMFlow/EEMFlow+.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_plus(modp, tag, seed, batch, h, w, cin, keep):
    from eemflow_amd.eemflow_plus import EEMFlow_cdc as Mirror
    from eemflow_amd.plus_weights import seeded_from_shapes
    shapes = {k: tuple(v.shape) for k, v in Mirror("", 3, cin).state_dict().items()}
    sd = seeded_from_shapes(shapes, seed)
    net = modp.EEMFlow_cdc(config="", groups=3, n_first_channels=cin).eval()
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 4000, batch, h, w, bins=cin))
    with torch.no_grad():
        (_, _), preds = net(e1, e2)
    arrays = dict(seed=np.int64(seed), input_seed=np.int64(seed + 4000), batch=np.int64(batch), hw=np.array([h, w]),
                  cin=np.int64(cin), pad=np.array(net.image_padder._pad), preds=torch.stack(preds).numpy())
    if keep:
        keys = list(net.state_dict().keys())
        shp = np.full((len(keys), 4), -1, dtype=np.int64)
        for i, v in enumerate(net.state_dict().values()):
            shp[i, :v.dim()] = list(v.shape)
        arrays.update(keys=np.array(keys), shapes=shp)
        # warp family on small tensors (cdc_utils.py:50-103, EEMFlow+.py:137-149)
        rng = np.random.default_rng(seed + 1)
        x = rng.standard_normal((2, 4, 9, 11), dtype=np.float32)
        flo = (rng.standard_normal((2, 2, 9, 11)) * 3).astype(np.float32)
        from model.EEMFlow.cdc_utils import WarpingLayer_no_div, upsample2d_flow_as
        tt = sys.modules["utils_luo.tools"].tensor_tools
        with torch.no_grad():
            small = torch.from_numpy((rng.standard_normal((2, 2, 5, 6)) * 2).astype(np.float32))
            small_in = small.clone()
            up = upsample2d_flow_as(small_in, torch.zeros(2, 1, 10, 12), mode="bilinear", if_rate=True)
            arrays.update(w_x=x, w_flo=flo, w_no_div=WarpingLayer_no_div()(torch.from_numpy(x), torch.from_numpy(flo)).numpy(),
                          w_torch_warp=tt.torch_warp(torch.from_numpy(x), torch.from_numpy(flo)).numpy(),
                          w_align_true=net.warp(torch.from_numpy(x), torch.from_numpy(flo)).numpy(),
                          up_in=small.numpy(), up_out=up.numpy(), up_in_after=small_in.numpy())
    save(f"eemflow_plus_{tag}.npz", **arrays)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    mod, lu, InputPadder = load_reference()
    gen_pad(InputPadder)
    gen_layout(mod)
    gen_corr(mod)
    gen_decoder(mod)
    gen_eemflow(mod, "128x192", seed=1, batch=2, h=128, w=192, keep_stages=True)
    gen_eemflow(mod, "260x346", seed=2, batch=1, h=260, w=346, keep_stages=False)
    gen_eemflow(mod, "100x150", seed=4, batch=1, h=100, w=150, keep_stages=False)
    gen_voxel(lu)
    gen_train(mod)
    modp = load_reference_plus()
    gen_plus(modp, "128x192", seed=12, batch=1, h=128, w=192, cin=5, keep=True)
    gen_plus(modp, "100x150_c15", seed=13, batch=2, h=100, w=150, cin=15, keep=False)
    gen_eraft_layout()
    gen_eraft_lookup()
    gen_eraft_upsample()
    # the reference's bilinear_sampler divides by (W-1), (H-1): the 1/64-scale pyramid level must be >= 2x2,
    # i.e. the padded input >= 128x128, or every output is NaN
    gen_eraft("128x160", seed=7, batch=1, h=128, w=160, iters=3, keep=True)
    gen_eraft("136x200", seed=8, batch=2, h=136, w=200, iters=2, keep=False)


if __name__ == "__main__":
    main()
