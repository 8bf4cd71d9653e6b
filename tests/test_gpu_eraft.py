"""E-RAFT on the GPU (through the C ABI) against reference-generated goldens and the oracle.  `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest
import torch

from eemflow_amd import _lib
from eemflow_amd.eraft import ERAFT
from eemflow_amd.eraft_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair
from oracle import eemflow_oracle as O
from oracle import eraft_oracle as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLOW_TOL = 1e-3      # north star tolerance on flow (12 recurrent fp32 iterations amplify reordering noise)
FEAT_TOL = 2e-4


def make_net(seed):
    net = ERAFT("", 5).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = seeded_from_shapes(shapes, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.keep_stages = True                     # the golden comparisons read the first iteration's intermediates
    return net.to(DEV), O.to_torch_sd(sd)


def maxerr(a, b):
    """max |a-b| over the positions where the reference b is finite (a 1x1 pyramid level divides by W-1 = 0 in
    the reference's bilinear_sampler and yields NaN there); a must be finite wherever b is."""
    a, b = torch.as_tensor(a).cpu().float(), torch.as_tensor(b).cpu().float()
    ok = torch.isfinite(b)
    assert bool(torch.isfinite(a[ok]).all())
    return float((a[ok] - b[ok]).abs().max())


def stream():
    return _lib.current_stream_ptr(torch.device(DEV))


def test_corr_lookup_golden(golden):
    g = golden("eraft_lookup.npz")
    net, _ = make_net(1)
    ctx = net._context(torch.device(DEV))
    f1, f2, co = (torch.from_numpy(g[k]).to(DEV) for k in ("f1", "f2", "coords"))
    b, c, h, w = f1.shape
    out = torch.empty(b, 324, h, w, device=DEV)
    _lib.check(_lib.lib().eraft_corr_lookup(ctx, f1.data_ptr(), f2.data_ptr(), co.data_ptr(), b, c, h, w, out.data_ptr(), stream()))
    for l in range(4):
        assert maxerr(net.stage(f"pyr{l}"), g[f"pyr{l}"]) < 1e-5
    assert maxerr(out, g["out"]) < 1e-5


def test_convex_upsample_golden(golden):
    g = golden("eraft_upsample.npz")
    net, _ = make_net(1)
    ctx = net._context(torch.device(DEV))
    flow, mask = torch.from_numpy(g["flow"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    b, _, h, w = flow.shape
    out = torch.empty(b, 2, 8 * h, 8 * w, device=DEV)
    _lib.check(_lib.lib().eraft_convex_upsample(ctx, flow.data_ptr(), mask.data_ptr(), b, h, w, out.data_ptr(), stream()))
    assert maxerr(out, g["up"]) < 1e-5


@pytest.mark.parametrize("tag", ["128x160", "136x200"])
def test_forward_vs_golden(golden, tag):
    g = golden(f"eraft_fwd_{tag}.npz")
    h, w = g["hw"].tolist()
    net, _ = make_net(int(g["seed"]))
    net.change_imagesize((h, w))
    assert net.image_padder._pad == g["pad"].tolist()
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w))
    with torch.no_grad():
        (r1, r2), preds = net(e1, e2, iters=int(g["iters"]))
    assert r1 is e1 and len(preds) == int(g["iters"]) and preds[0].shape == (int(g["batch"]), 2, h, w)
    b = int(g["batch"])
    if "fmap1" in g.files:
        fm = net.stage("fmap")
        assert maxerr(fm[:b], g["fmap1"]) < FEAT_TOL and maxerr(fm[b:], g["fmap2"]) < FEAT_TOL
        assert maxerr(net.stage("inp"), g["inp"]) < FEAT_TOL
        assert maxerr(net.stage("pyr1"), g["pyr1"]) < FEAT_TOL and maxerr(net.stage("pyr3"), g["pyr3"]) < FEAT_TOL
        assert maxerr(net.stage("corr0"), g["corr0"]) < FEAT_TOL
        assert maxerr(net.stage("net1"), g["net1"]) < FEAT_TOL
        assert maxerr(net.stage("delta1"), g["delta1"]) < FEAT_TOL
        assert maxerr(net.stage("mask1"), g["mask1"]) < FEAT_TOL
    assert maxerr(torch.stack(preds), g["preds"]) < FLOW_TOL


@pytest.mark.parametrize("b,h,w,iters", [(1, 128, 160, 2), (3, 136, 200, 4), (3, 256, 352, 2)])
def test_forward_vs_oracle(b, h, w, iters):
    """The last case is large enough (32x44 grid, batch 3) for the LDS-tiled conv kernel and ragged in its tiles."""
    net, sd = make_net(17)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(18, b, h, w))
    with torch.no_grad():
        preds = net(e1.to(DEV), e2.to(DEV), iters=iters)[1]
        ref, _ = R.eraft_forward(sd, e1, e2, iters=iters)
    assert maxerr(torch.stack(preds), torch.stack(ref)) < FLOW_TOL


def test_split_k_conv_launches_equal_plain_ones(monkeypatch):
    """The split-K form of the generic conv (one tile per block, k-batches round robin over the waves; taken by the small
    launches of the 1/8-resolution update block) against the plain form (EEM_NO_SPLITK=1, read at launch)."""
    h, w = 480, 640
    net, _ = make_net(21)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(22, 1, h, w))
    with torch.no_grad():
        fast = torch.stack(net(e1, e2, iters=3)[1]).clone()
        monkeypatch.setenv("EEM_NO_SPLITK", "1")
        plain = torch.stack(net(e1, e2, iters=3)[1])
    assert maxerr(fast, plain) < 2e-4 and float(plain.abs().max()) > 1e-3


def test_lds_tiled_conv_equals_generic_conv(monkeypatch):
    """gconv16 (LDS-tiled 16x16x4 path of the stride-1, 16-aligned convs) against the generic kernel (EEM_NO_GCONV16=1, read
    at launch) through a whole forward at batch 2: same flow to fp32 summation-order round-off."""
    h, w = 480, 640
    net, _ = make_net(27)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(28, 2, h, w))
    with torch.no_grad():
        fast = torch.stack(net(e1, e2, iters=3)[1]).clone()
        monkeypatch.setenv("EEM_NO_GCONV16", "1")
        plain = torch.stack(net(e1, e2, iters=3)[1])
    assert maxerr(fast, plain) < 2e-4 and float(plain.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w,iters", [(2, 256, 352, 3), (1, 480, 640, 3), (4, 480, 640, 2), (1, 136, 200, 2)])
def test_bf16_piece_convs_equal_the_fp32_ones(monkeypatch, b, h, w, iters):
    """gconvb.hip (the 32-aligned stride-1 convs on the bf16 matrix pipe: exact three-piece operands, six MFMAs per product, 128-pixel x
    64-cout tiles; default for launches of >= 256 blocks, EEM_GCONVB_MINBLK=1 sends every eligible launch through it - both read per
    call) against the fp32-MFMA kernels (EEM_NO_GCONVB=1): every epilogue the update block and the encoders use - GRU blend, z | r with
    r * h, residual add + ReLU, the per-pixel context addend - and every kernel shape (1x1, 3x3, 1x5, 5x1; one to three input segments)."""
    net, _ = make_net(29)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(30, b, h, w))
    with torch.no_grad():
        monkeypatch.setenv("EEM_GCONVB_MINBLK", "1")
        monkeypatch.setenv("EEM_GCONVB_1X1", "1")
        fast = torch.stack(net(e1, e2, iters=iters)[1]).clone()
        monkeypatch.setenv("EEM_NO_GCONVB", "1")
        plain = torch.stack(net(e1, e2, iters=iters)[1])
    assert not torch.equal(fast, plain)                       # (the switch did switch)
    assert maxerr(fast, plain) < 2e-4 and float(plain.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w,iters", [(2, 256, 352, 3), (1, 480, 640, 12), (1, 136, 200, 2)])
def test_on_the_fly_correlation_equals_the_resident_volume(b, h, w, iters):
    """ERAFT.alternate_corr (eraft_set_alternate_corr; the alt_cuda_corr pattern): the 324 correlation features from
    <fmap1, pooled fmap2> over each pixel's 10 x 10 window instead of from the all-pairs volume - the same first lookup to summation
    order, the same flow within the flow tolerance after 12 iterations, and the oracle's flow."""
    net, sd = make_net(33)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(34, b, h, w))
    with torch.no_grad():
        vol = torch.stack(net(e1.to(DEV), e2.to(DEV), iters=iters)[1]).clone()
        corr_vol = net.stage("corr0").clone()
        net.alternate_corr = True
        fly = torch.stack(net(e1.to(DEV), e2.to(DEV), iters=iters)[1]).clone()
        corr_fly = net.stage("corr0").clone()
        with pytest.raises(_lib.EEMFlowHipError, match="on the fly"):
            net.stage("pyr0")
    assert maxerr(corr_fly, corr_vol) < 2e-5 * max(1.0, float(corr_vol.abs().max()))
    assert maxerr(fly, vol) < (FLOW_TOL if iters > 3 else 1e-4) and float(vol.abs().max()) > 1e-3
    if h * w <= 256 * 352:
        ref, _ = R.eraft_forward(sd, e1, e2, iters=iters)
        assert maxerr(fly[-1], ref[-1]) < FLOW_TOL


@pytest.mark.parametrize("b,h,w,iters,keep", [(2, 136, 200, 5, False), (1, 480, 640, 12, False), (4, 256, 352, 4, False),
                                              (1, 136, 200, 1, False), (2, 136, 200, 3, True)])
def test_final_only_is_the_last_of_the_full_list(b, h, w, iters, keep):
    """ERAFT.final_only (eraft_set_final_only): what the evaluation loop reads (test_mvsec.py:1455 flow_list[-1]) without the mask
    head and the convex upsampling of the iterations before the last - the one returned prediction and the low-resolution flow are
    bit for bit those of the full forward (odd and even iteration counts: coords1 stays in one buffer until the last iteration); with
    keep_stages the first iteration still forms its mask; the switch can be turned back."""
    net, _ = make_net(71)
    net.keep_stages = keep
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(72, b, h, w))
    with torch.no_grad():
        full = torch.stack(net(e1, e2, iters=iters)[1]).clone()
        low_full = net.stage("flow_low").clone()
        mask_full = net.stage("mask1").clone() if keep else None
        net.final_only = True
        one = net(e1, e2, iters=iters)[1]
        low_one = net.stage("flow_low").clone()
        assert len(one) == 1 and one[0].shape == full[-1].shape
        last = one[0].clone()
        if keep:
            assert torch.equal(net.stage("mask1"), mask_full)
        net.final_only = False
        again = torch.stack(net(e1, e2, iters=iters)[1])
    assert torch.equal(last, full[-1]) and torch.equal(low_one, low_full)
    assert torch.equal(again, full) and float(full.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w,iters,final", [(2, 136, 200, 5, False), (1, 480, 640, 12, False), (4, 480, 640, 3, False),
                                               (1, 480, 640, 4, True), (1, 64, 96, 3, False)])
def test_mask_head_beside_the_next_iteration_equals_the_serial_order(monkeypatch, b, h, w, iters, final):
    """The default schedule - mask head and convex upsampling of iteration i on the side stream behind the flow branch of iteration
    i + 1, the flow head's last conv writing delta_flow and coords1 + delta_flow in one launch - against the mask head inside the iteration
    (EEM_ERAFT_NO_LAG=1, read per forward): every prediction, the low-resolution flow and the kept stages of the first iteration are
    the same (the hidden state is read by the lagging launches while the next iteration reads it, and overwritten only after they have
    finished); with final_only; on a map of less than 256 cells (the generic kernel and a separate sum); twice in a row on one
    context (the events are reused)."""
    net, _ = make_net(91)
    net.final_only = final
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(92, b, h, w))
    outs = {}
    for lag in ("1", "0", "0"):
        monkeypatch.setenv("EEM_ERAFT_NO_LAG", lag)
        with torch.no_grad():
            preds = torch.stack(net(e1, e2, iters=iters)[1]).clone()
            outs[lag] = (preds, net.stage("flow_low").clone(), net.stage("mask1").clone(), net.stage("delta1").clone(), net.stage("net1").clone())
    for x, y in zip(outs["0"], outs["1"]):
        assert maxerr(x, y) < 1e-5 * max(1.0, float(y.abs().max()))
    assert float(outs["1"][0].abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w,iters", [(1, 480, 640, 2), (4, 480, 640, 1), (3, 136, 200, 2), (1, 128, 160, 2)])
def test_stem_kernel_equals_the_taps_kernel(monkeypatch, b, h, w, iters):
    """conv_stem7.hip (the encoders' 7x7 stride-2 stem as an implicit GEMM over (channel, ky, kx): the patch and the weights in LDS,
    16x16x4 MFMAs) against gconv_taps_kernel<7, 7> (EEM_NO_STEM7=1, read per call): the feature maps of both networks agree to
    summation order, on full tiles and on ragged ones (112 and 80 output columns are not multiples of 32), with folded BatchNorm + ReLU
    (context network) and plain (feature network) epilogues."""
    net, _ = make_net(97)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(98, b, h, w))
    with torch.no_grad():
        new = torch.stack(net(e1, e2, iters=iters)[1]).clone()
        fm_new, inp_new = net.stage("fmap").clone(), net.stage("inp").clone()
        monkeypatch.setenv("EEM_NO_STEM7", "1")
        old = torch.stack(net(e1, e2, iters=iters)[1])
        fm_old, inp_old = net.stage("fmap").clone(), net.stage("inp").clone()
    assert not torch.equal(fm_new, fm_old)                    # (the switch did switch)
    assert maxerr(fm_new, fm_old) < 2e-5 * max(1.0, float(fm_old.abs().max())) and maxerr(inp_new, inp_old) < 2e-5 * max(1.0, float(inp_old.abs().max()))
    assert maxerr(new, old) < 2e-4 and float(old.abs().max()) > 1e-3


@pytest.mark.parametrize("n,h,w,iters,final", [(3, 136, 200, 4, False), (4, 480, 640, 3, False), (2, 256, 352, 5, True), (1, 128, 160, 2, False)])
def test_forward_many_equals_the_batched_forward(n, h, w, iters, final):
    """ERAFT.forward_many (eraft_forward_many): n independent batch-1 samples in their own tensors through one batch-n forward - every
    prediction bitwise what `forward` returns for the samples stacked into a batch (padded sizes, `final_only`); argument errors for an
    empty list and for batched frames."""
    net, _ = make_net(101)
    net.final_only = final
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(102, n, h, w))
    with torch.no_grad():
        batched = torch.stack(net(e1, e2, iters=iters)[1]).clone()          # [iters or 1, n, 2, h, w]
        frames = [(e1[i:i + 1].clone(), e2[i:i + 1].clone()) for i in range(n)]
        many = net.forward_many(frames, iters=iters)
    assert len(many) == n
    for i, ((a, b), preds) in enumerate(many):
        assert a is frames[i][0] and len(preds) == (1 if final else iters) and preds[0].shape == (1, 2, h, w)
        assert torch.equal(torch.stack(preds)[:, 0], batched[:, i])
    assert float(batched.abs().max()) > 1e-3
    with pytest.raises(ValueError):
        net.forward_many([])
    with pytest.raises(ValueError):
        net.forward_many([(torch.cat([e1[:1], e1[:1]]), torch.cat([e2[:1], e2[:1]]))])


@pytest.mark.parametrize("b,h,w,iters", [(1, 480, 640, 3), (2, 136, 200, 4)])
def test_one_stream_forward_equals_the_three_stream_one(monkeypatch, b, h, w, iters):
    """EEM_ERAFT_NO_OVERLAP=1 (read per forward): every launch on the caller's stream - no side stream for the context network, the
    flow branch and the lagging mask head - against the default schedule: the same kernels on the same operands in
    another order of enqueueing, predictions and kept stages equal to the narrower launches' summation order."""
    net, _ = make_net(105)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(106, b, h, w))
    with torch.no_grad():
        three = torch.stack(net(e1, e2, iters=iters)[1]).clone()
        st3 = [net.stage(k).clone() for k in ("flow_low", "net1", "mask1", "delta1")]
        monkeypatch.setenv("EEM_ERAFT_NO_OVERLAP", "1")
        one = torch.stack(net(e1, e2, iters=iters)[1]).clone()
        st1 = [net.stage(k).clone() for k in ("flow_low", "net1", "mask1", "delta1")]
    assert maxerr(three, one) < 2e-4 and float(one.abs().max()) > 1e-3
    for x, y in zip(st3, st1):
        assert maxerr(x, y) < 2e-5 * max(1.0, float(y.abs().max()))


def test_frames_in_flight_hint_changes_tiles_not_results():
    """eraft_set_frames_in_flight >= 3: the LDS-tiled convs of 512..2047 blocks use 4-row tiles instead of 2-row ones (batch 4 at
    640x480: the update block), and launches of at most one block per CU keep one group of waves per tile instead of two that split
    the channel chunks (a different summation order): the flow changes by round-off only."""
    h, w = 480, 640
    net, _ = make_net(29)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(30, 4, h, w))
    with torch.no_grad():
        one = torch.stack(net(e1, e2, iters=3)[1]).clone()
        net.frames_in_flight = 4
        many = torch.stack(net(e1, e2, iters=3)[1]).clone()
    assert maxerr(many, one) < 2e-4 and float(one.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w,iters", [(2, 136, 200, 5), (1, 480, 640, 4)])
def test_fused_small_launches_equal_the_separate_ones(monkeypatch, b, h, w, iters):
    """flow = coords1 - coords0 written by the lookup launch and coords1 += delta_flow carried by the convex-upsampling launch
    (coords1 alternating between two buffers) against the separate flow / axpy launches (EEM_ERAFT_NO_FUSE=1, read per forward): the
    same additions and subtractions in the same order - every prediction bitwise the same, at a padded size (cells of the padding
    have no output pixel) and for odd and even iteration counts; the low-resolution flow (coords1 of the last iteration) too."""
    net, _ = make_net(61)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(62, b, h, w))
    outs = []
    for off in ("0", "1"):
        monkeypatch.setenv("EEM_ERAFT_NO_FUSE", off)
        with torch.no_grad():
            preds = torch.stack(net(e1, e2, iters=iters)[1]).clone()
            low = net.stage("flow_low").clone()
        outs.append((preds, low))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 1e-3


def test_flow_init_and_twelve_iterations():
    h, w = 128, 128
    net, sd = make_net(19)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(20, 1, h, w))
    init = torch.randn(1, 2, 16, 16, generator=torch.Generator().manual_seed(2)) * 0.5
    with torch.no_grad():
        preds = net(e1.to(DEV), e2.to(DEV), iters=12, flow_init=init.to(DEV))[1]
        ref, _ = R.eraft_forward(sd, e1, e2, iters=12, flow_init=init)
    assert len(preds) == 12 and maxerr(preds[-1], ref[-1]) < FLOW_TOL


@pytest.mark.parametrize("norm_fn", ["none", "instance", "batch", "group"])
def test_basic_encoder_standalone_forward(norm_fn):
    """BasicEncoder(output_dim, norm_fn, dropout, n_first_channels).forward (model/extractor.py:119-190) on its own - the four norm
    settings of the reference ('none' = its empty nn.Sequential, 'group' = GroupNorm(planes // 8) with random affine parameters here), a
    list input run as one batch and split again - against the oracle, with gradients flowing (the operator-level autograd route)."""
    from eemflow_amd.eraft import BasicEncoder
    torch.manual_seed(5)
    enc = BasicEncoder(output_dim=96, norm_fn=norm_fn, dropout=0.0, n_first_channels=5).to(DEV).eval()
    if norm_fn == "group":
        with torch.no_grad():
            for m in enc.modules():
                if isinstance(m, torch.nn.GroupNorm):
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    sd = {k: v.detach().cpu().float() for k, v in enc.state_dict().items()}
    a, b = (torch.from_numpy(x) for x in synthetic_voxel_pair(9, 2, 128, 160))
    ya, yb = enc([a.to(DEV), b.to(DEV)])
    ra, rb = R.basic_encoder(sd, "", [a, b], norm_fn)
    assert ya.shape == (2, 96, 16, 20)
    scale = float(ra.abs().max())
    assert maxerr(ya, ra) < 2e-4 * max(scale, 1.0) and maxerr(yb, rb) < 2e-4 * max(scale, 1.0)
    single = enc(a.to(DEV))
    assert maxerr(single, R.basic_encoder(sd, "", a, norm_fn)) < 2e-4 * max(scale, 1.0)
    single.square().mean().backward()
    assert enc.conv1.weight.grad is not None and float(enc.conv1.weight.grad.abs().max()) > 0
    if norm_fn == "group":                                       # the affine parameters train: gradient against torch autograd through the oracle
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        R.basic_encoder(params, "", a, norm_fn).square().mean().backward()
        for name in ("norm1.weight", "norm1.bias", "layer2.0.norm3.weight", "layer3.1.norm2.bias"):
            got = dict(enc.named_parameters())[name].grad.cpu()
            ref = params[name].grad
            assert maxerr(got, ref) < 3e-3 * max(float(ref.abs().max()), 1e-6), name
    with pytest.raises(ValueError):
        BasicEncoder(norm_fn="layer")


def test_basic_encoder_dropout_in_training_mode():
    """dropout > 0 (model/extractor.py:147-149,183-184): nn.Dropout2d on the output in train() - whole channels zero with probability p,
    the others scaled by 1 / (1 - p); identity in eval()."""
    from eemflow_amd.eraft import BasicEncoder
    torch.manual_seed(6)
    enc = BasicEncoder(output_dim=64, norm_fn="instance", dropout=0.5, n_first_channels=5).to(DEV)
    a = torch.from_numpy(synthetic_voxel_pair(9, 2, 64, 96)[0]).to(DEV)
    enc.eval()
    ref = enc(a).detach()
    enc.train()
    y = enc(a)
    dropped = (y.abs().amax(dim=(2, 3)) == 0)
    assert 0 < int(dropped.sum()) < dropped.numel()
    kept = ~dropped
    assert maxerr((y.detach() * kept[:, :, None, None])[kept], (2.0 * ref * kept[:, :, None, None])[kept]) < 1e-5 * max(1.0, float(ref.abs().max()))
    y.sum().backward()
    assert float(enc.conv2.weight.grad.abs().max()) > 0


def test_errors():
    net, _ = make_net(1)
    with pytest.raises(AttributeError):
        net(torch.zeros(1, 5, 64, 64, device=DEV), torch.zeros(1, 5, 64, 64, device=DEV))
    net.change_imagesize((64, 64))
    with pytest.raises(_lib.EEMFlowHipError):
        net(torch.zeros(1, 5, 64, 64), torch.zeros(1, 5, 64, 64))


def test_encoder_winograd_convs_equal_the_lds_tiled_ones(monkeypatch):
    """The encoder's 64 -> 64 residual convs (model/extractor.py layer1, both networks) run on EEMFlow's Winograd F(4x4,3x3) kernel from
    128 tiles on - an eval-mode BatchNorm's scale folded into the weights, ReLU and the residual sum in its epilogue (cnet), no
    activation in front of the InstanceNorm (fnet); EEM_ERAFT_NO_F4=1 (read per forward) keeps them on gconv16.  Same sums in another
    order: the twelve predictions agree far inside the 1e-3 budget."""
    h, w = 480, 640
    net, _ = make_net(43)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(44, 1, h, w))
    with torch.no_grad():
        wino = torch.stack(net(e1, e2, iters=12)[1]).clone()
        monkeypatch.setenv("EEM_ERAFT_NO_F4", "1")
        tiled = torch.stack(net(e1, e2, iters=12)[1]).clone()
    assert not torch.equal(wino, tiled)                                  # (the switch did switch)
    # F(4x4) differs from a direct convolution by ~1e-5 of a feature's range; twelve recurrent iterations carry that to ~2e-4 px
    assert maxerr(wino, tiled) < 5e-4 and float(tiled.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w", [(1, 480, 640), (2, 256, 320)])
def test_encoder_winograd_f2_convs_equal_the_bf16_piece_ones(monkeypatch, b, h, w):
    """The encoder's 96 -> 96 and 128 -> 128 residual convs (model/extractor.py layer2 / layer3, both networks) run on the Winograd
    F(2x2,3x3) kernel of EEMFlow+'s fine levels (conv_wnc.hip: their 32-cout slices as the jobs of one launch, BatchNorm scale folded
    into the weights, ReLU and the residual sum in the epilogue for cnet, no activation in front of fnet's InstanceNorm) from 128
    (tile, slice) pairs on, and so does the motion encoder's convf2 128 -> 64 (model/update.py:69; EEM_ERAFT_WNC_UPD masks it);
    EEM_ERAFT_NO_WNC=1 (read per call) keeps all of them on the bf16-piece / LDS-tiled kernels."""
    net, _ = make_net(47)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(48, b, h, w))
    with torch.no_grad():
        wino = torch.stack(net(e1, e2, iters=12)[1]).clone()
        monkeypatch.setenv("EEM_ERAFT_NO_WNC", "1")
        plain = torch.stack(net(e1, e2, iters=12)[1]).clone()
    assert not torch.equal(wino, plain)                                  # (the switch did switch)
    assert maxerr(wino, plain) < 5e-4 and float(plain.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w", [(1, 480, 640), (2, 200, 296)])
def test_all_pairs_volume_through_lds_equals_the_l2_form(monkeypatch, b, h, w):
    """The all-pairs correlation (model/corr.py:53-60) with both feature maps staged through LDS in chunks of 16 channels against the form
    that feeds every MFMA from L2 (EEM_ALLPAIRS_L2=1, read per call): the same products summed in the same order - the same twelve
    predictions bit for bit (60x80 = 37.5 block tiles: pixels past the last read a zero page; 25x37 cells are not a multiple of four: both
    runs take the L2 form there and the test only says so)."""
    net, _ = make_net(49)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(50, b, h, w))
    with torch.no_grad():
        lds = torch.stack(net(e1, e2, iters=12)[1]).clone()
        monkeypatch.setenv("EEM_ALLPAIRS_L2", "1")
        l2 = torch.stack(net(e1, e2, iters=12)[1]).clone()
    assert torch.equal(lds, l2) and float(l2.abs().max()) > 1e-3


@pytest.mark.parametrize("b,h,w", [(1, 480, 640), (1, 200, 296), (3, 256, 320)])
def test_lookup_blocks_of_16_pixels_and_the_one_launch_pyramid_equal_the_round5_forms(monkeypatch, b, h, w):
    """Two launch shapes of `CorrBlock` (model/corr.py:24-50) that compute the same values: the lookup with 16 or 32 pixels per block
    instead of 64 (EEM_LOOKUP_PX, read per call; the default below 1 024 blocks of 64), and levels 1 .. 3 of the pyramid from one launch
    that reads level 0 once (EEM_POOL_CHAIN=1: three avg_pool2d launches).  Same samples, same expressions: the twelve predictions bit
    for bit (25x37 cells: odd maps, a level-3 map of 3x4, a last band of one level-1 row)."""
    net, _ = make_net(51)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(52, b, h, w))
    with torch.no_grad():
        now = torch.stack(net(e1, e2, iters=12)[1]).clone()
        for px in ("16", "32", "64"):
            monkeypatch.setenv("EEM_LOOKUP_PX", px)
            assert torch.equal(torch.stack(net(e1, e2, iters=12)[1]), now), px
        monkeypatch.setenv("EEM_POOL_CHAIN", "1")
        old = torch.stack(net(e1, e2, iters=12)[1]).clone()
    assert torch.equal(now, old) and float(old.abs().max()) > 1e-3


def test_flow_head_on_32_pixel_blocks_equals_the_64_pixel_form(monkeypatch):
    """The flow head's 256 -> 2 conv (model/update.py:6-13) at batch 1: blocks of 32 pixels x 32 channel groups with all of a lane's taps
    in registers against 64 pixels x 16 groups of request rounds (EEM_FEWOUT_WIDE=0, read per call).  The same products, 32 partial sums
    instead of 16: equal within rounding over twelve iterations, not bitwise."""
    h, w = 480, 640
    net, _ = make_net(53)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(54, 1, h, w))
    with torch.no_grad():
        wide = torch.stack(net(e1, e2, iters=12)[1]).clone()
        monkeypatch.setenv("EEM_FEWOUT_WIDE", "0")
        old = torch.stack(net(e1, e2, iters=12)[1]).clone()
    assert not torch.equal(wide, old)                                    # (the switch did switch)
    assert maxerr(wide, old) < 2e-4 and float(old.abs().max()) > 1e-3


def test_gru_context_part_computed_once_equals_the_full_convs(monkeypatch):
    """The GRU's convs read [h | inp | motion] (model/update.py:43-60) and `inp` does not change over the iterations: its part of
    every conv (+ the bias) is computed once per forward and enters the in-loop convs over [h | motion] as a per-pixel addend in front
    of the activation (GConvArgs::pre); EEM_ERAFT_NO_PRE=1 (read per forward) keeps the reference's full convolutions.  The same
    products summed in another order."""
    h, w = 256, 320
    net, _ = make_net(45)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(46, 2, h, w))
    with torch.no_grad():
        once = torch.stack(net(e1, e2, iters=6)[1]).clone()
        monkeypatch.setenv("EEM_ERAFT_NO_PRE", "1")
        full = torch.stack(net(e1, e2, iters=6)[1]).clone()
    assert not torch.equal(once, full)                                   # (the switch did switch)
    assert maxerr(once, full) < 2e-4 and float(full.abs().max()) > 1e-3


def test_r_times_h_in_the_gate_conv_epilogue_equals_the_separate_launch(monkeypatch):
    """z | r of a GRU pass is one conv; its r half leaves the launch as r * h (GEPI_ZR, second output) instead of through an elementwise
    launch (EEM_ERAFT_NO_ZR=1, read per forward).  The same products: bitwise equal."""
    h, w = 256, 320
    net, _ = make_net(47)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(48, 2, h, w))
    with torch.no_grad():
        fused = torch.stack(net(e1, e2, iters=4)[1]).clone()
        monkeypatch.setenv("EEM_ERAFT_NO_ZR", "1")
        apart = torch.stack(net(e1, e2, iters=4)[1]).clone()
    assert torch.equal(fused, apart) and float(apart.abs().max()) > 1e-3


def test_stacked_update_block_launches_equal_separate_ones(monkeypatch):
    """z | r of a GRU pass and flow-head | mask-head conv1 run as one launch each, stacked along the output channels
    (csrc/eraft_api.hip); EEM_ERAFT_NO_STACK=1 (read per forward) keeps the reference's five separate convolutions.  Per output
    channel the same kernel, tile shape permitting the same k order: equal to round-off."""
    h, w = 256, 320
    net, _ = make_net(41)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(42, 2, h, w))
    with torch.no_grad():
        stacked = torch.stack(net(e1, e2, iters=4)[1]).clone()
        monkeypatch.setenv("EEM_ERAFT_NO_STACK", "1")
        apart = torch.stack(net(e1, e2, iters=4)[1]).clone()
    assert maxerr(stacked, apart) < 2e-4 and float(apart.abs().max()) > 1e-3
