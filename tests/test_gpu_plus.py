"""EEMFlow+ (EEMFlow_cdc) on the GPU against reference-generated goldens.  `pytest -m gpu`.

The reference's WarpingLayer_no_div masks with `grid_sample(ones) >= 1.0`; that test flips on 1-ulp changes of
the flow, so past the first warped level the REFERENCE ITSELF moves by up to ~0.1 px on a third of the pixels
when only its CPU thread count changes (tests/test_oracle_golden.py::test_plus_forward).  Hence: the warp /
resampling ops are checked bit-for-bit on fixed inputs; every level is held to the 1e-3 flow tolerance TEACHER-FORCED
(from the reference's own flow_init of that level, and from the GPU chain's own flow_init against the oracle); only
the end-to-end chained comparison keeps the robust statistics the reference satisfies against itself."""
import ctypes

import numpy as np
import pytest
import torch

from eemflow_amd import _lib
from eemflow_amd.eemflow_plus import EEMFlow_cdc
from eemflow_amd.plus_weights import seeded_from_shapes
from eemflow_amd.weights import synthetic_voxel_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def stream():
    return _lib.current_stream_ptr(torch.device(DEV))


def make_net(seed, cin):
    net = EEMFlow_cdc("", 3, cin).eval()
    sd = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(DEV)


@pytest.mark.parametrize("mode,key", [(2, "w_no_div"), (1, "w_torch_warp"), (0, "w_align_true")])
def test_warp_family_same_mask(golden, mode, key):
    g = golden("eemflow_plus_128x192.npz")
    x, flo = torch.from_numpy(g["w_x"]).to(DEV), torch.from_numpy(g["w_flo"]).to(DEV)
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().eemplus_warp(x.data_ptr(), flo.data_ptr(), b, c, h, w, mode, out.data_ptr(), stream()))
    got, ref = out.cpu().numpy(), g[key]
    assert np.array_equal(got == 0, ref == 0)                  # the >= 1.0 mask and the zero padding: identical support
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6)    # values: few-ulp differences in the 4-tap sum only


def test_upsample_flow_as_and_its_side_effect(golden):
    g = golden("eemflow_plus_128x192.npz")
    inp = torch.from_numpy(g["up_in"]).to(DEV).clone()
    b, _, h, w = inp.shape
    out = torch.empty(b, 2, 10, 12, device=DEV)
    _lib.check(_lib.lib().eemplus_upsample_flow_as(inp.data_ptr(), b, h, w, 10, 12, 1, out.data_ptr(), stream()))
    np.testing.assert_allclose(out.cpu().numpy(), g["up_out"], atol=1e-5)
    np.testing.assert_allclose(inp.cpu().numpy(), g["up_in_after"], rtol=1e-7)     # inputs scaled in place (cdc_utils.py:85-86)


@pytest.mark.parametrize("tag", ["128x192", "100x150_c15"])
def test_forward_vs_golden(golden, tag):
    g = golden(f"eemflow_plus_{tag}.npz")
    h, w = g["hw"].tolist()
    cin, b = int(g["cin"]), int(g["batch"])
    net = make_net(int(g["seed"]), cin)
    assert len(net.state_dict()) == 136
    net.change_imagesize((h, w))
    assert net.image_padder._pad == g["pad"].tolist()
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(int(g["input_seed"]), b, h, w, bins=cin))
    with torch.no_grad():
        (r1, _), preds = net(e1, e2)
    assert r1 is e1 and len(preds) == 5 and preds[0].shape == (b, 2, h, w)
    got, ref = torch.stack(preds).cpu().numpy(), g["preds"]
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    assert err[0].max() < 1e-3                                  # level 6: no warp upstream
    for i in range(1, 5):                                       # levels 5..2: mask flips allowed, bounded like the reference's own
        e = err[i].ravel()
        assert np.median(e) < 2e-3, (i, float(np.median(e)))
        assert (e > 0.25).mean() < 0.02, (i, float((e > 0.25).mean()))
        assert e.max() < 2.0, (i, float(e.max()))


FLOW_TOL = 1e-3      # north star tolerance on flow


@pytest.mark.parametrize("tag", ["128x192", "100x150_c15"])
def test_levels_teacher_forced_vs_reference_golden(golden, tag):
    """A8 held to the flow tolerance level by level: every l-block of EEMFlow+.py:183-229 runs on the GPU from the REFERENCE's own
    flow_init of that level (hooks on the reference modules, tests/golden/make_golden_plus_levels.py) and must reproduce the
    reference's flow_up_l and flow_l; the upsampling that links the levels (and its in-place doubling) is checked separately."""
    g = golden(f"eemflow_plus_levels_{tag}.npz")
    h, w = g["hw"].tolist()
    cin, b = int(g["cin"]), int(g["batch"])
    net = make_net(int(g["seed"]), cin)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(int(g["input_seed"]), b, h, w, bins=cin))
    with torch.no_grad():
        net(e1, e2)                                              # builds the feature pyramid the levels run on
    assert float((net.stage("flow6").cpu() / 2 - torch.from_numpy(g["flow6"])).abs().max()) < FLOW_TOL   # stage carries the doubling
    for l in (5, 4, 3, 2):
        fin = torch.from_numpy(g[f"flow_in{l}"]).to(DEV).clone()
        hh, ww = g[f"flow_init{l}"].shape[-2:]
        init = torch.empty(b, 2, hh, ww, device=DEV)
        _lib.check(_lib.lib().eemplus_upsample_flow_as(fin.data_ptr(), b, fin.shape[2], fin.shape[3], hh, ww, 1, init.data_ptr(), stream()))
        assert float((init.cpu() - torch.from_numpy(g[f"flow_init{l}"])).abs().max()) < 1e-5, l
        np.testing.assert_allclose(fin.cpu().numpy(), 2 * g[f"flow_in{l}"], rtol=1e-6)
        up, fl = net.level(l, torch.from_numpy(g[f"flow_init{l}"]).to(DEV))
        e_up = float((up.cpu() - torch.from_numpy(g[f"flow_up{l}"])).abs().max())
        e_fl = float((fl.cpu() - torch.from_numpy(g[f"flow{l}"])).abs().max())
        assert e_up < FLOW_TOL and e_fl < FLOW_TOL, (l, e_up, e_fl)


@pytest.mark.parametrize("b,h,w,cin", [(1, 256, 320, 5), (2, 200, 300, 15), (1, 720, 1280, 5)])
def test_chained_forward_levels_vs_oracle_on_its_own_flow_init(b, h, w, cin):
    """The chained GPU forward, level by level, against the oracle: the oracle's l-block is fed the GPU's OWN flow_init of that level
    (stage "flow_init<l>"), so the comparison holds the flow tolerance at every level of the actual forward - the only thing not
    compared is the discontinuity of the mask w.r.t. its input, which belongs to the reference function itself."""
    from oracle import eemflow_oracle as O
    from oracle import eemflow_plus_oracle as P
    net = EEMFlow_cdc("", 3, cin).eval()
    sdn = seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 91)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()})
    net = net.to(DEV)
    sd = O.to_torch_sd(sdn)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(92, b, h, w, bins=cin))
    with torch.no_grad():
        net(e1.to(DEV), e2.to(DEV))
        _, st = P.eemflow_plus_forward(sd, e1, e2, keep=True)
        assert float((net.stage("flow6").cpu() - st["flow6"]).abs().max()) < FLOW_TOL          # level 6: no warp upstream
        for l in (5, 4, 3, 2):
            init = net.stage(f"flow_init{l}")
            up_gpu = net.stage(f"flow_up{l}")
            up2, fl_gpu = net.level(l, init)                     # same inputs, same kernels: the chain's own values
            assert torch.equal(up2, up_gpu)
            up_ref, fl_ref = P.level_from_init(sd, l, st["f1"][l], st["f2"][l], init.cpu())
            d_up, d_fl = (up_gpu.cpu() - up_ref).abs(), (fl_gpu.cpu() - fl_ref).abs()
            e_up, e_fl = float(d_up.max()), float(d_fl.max())
            if h * w < 512 * 512:
                assert e_up < FLOW_TOL and e_fl < FLOW_TOL, (l, e_up, e_fl)
            else:
                # 1280x720: of a level's 14 400 / 57 600 pixels one or two sit on the warp mask's `>= 1.0` threshold to the last bit of
                # ATen's vectorised CPU arithmetic (tools/plus720_probe.py: the same two pixels whichever conv kernels run); everything
                # else holds the tolerance, and those stay within a few 1e-3
                assert float((d_up > FLOW_TOL).float().mean()) < 5e-4 and float((d_fl > FLOW_TOL).float().mean()) < 5e-4, (l, e_up, e_fl)
                assert e_up < 2e-2 and e_fl < 2e-2, (l, e_up, e_fl)


@pytest.mark.parametrize("b,h,w,cin", [(1, 256, 320, 5), (1, 720, 1280, 5)])
def test_bf16_piece_convs_equal_the_fp32_ones(monkeypatch, b, h, w, cin):
    """EEMFlow+'s encoder / decoder convs through gconvb.hip (exact three-piece operands on the bf16 matrix pipe; EEM_GCONVB_MINBLK=1:
    every eligible launch) against the fp32-MFMA kernels (EEM_NO_GCONVB=1); both switches are read per call."""
    net = make_net(31, cin)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(32, b, h, w, bins=cin))
    with torch.no_grad():
        monkeypatch.setenv("EEM_GCONVB_MINBLK", "1")
        fast = net(e1, e2)[1][0].clone()
        monkeypatch.setenv("EEM_NO_GCONVB", "1")
        plain = net(e1, e2)[1][0]
    # (no "the switch did switch" assertion here: with seeded weights the flow is tens of pixels and the fine levels' decoders add corrections
    # whose round-off differences fall below one ulp of it; E-RAFT's test of the same kernels sees them)
    assert float((fast - plain).abs().max()) < 2e-4 and float(plain.abs().max()) > 1e-3


def test_errors():
    net = make_net(1, 5)
    with pytest.raises(AttributeError):
        net(torch.zeros(1, 5, 64, 64, device=DEV), torch.zeros(1, 5, 64, 64, device=DEV))
    net.change_imagesize((64, 64))
    with pytest.raises(_lib.EEMFlowHipError):
        net(torch.zeros(1, 5, 64, 64), torch.zeros(1, 5, 64, 64))


def test_fast_encoder_equals_generic_encoder(monkeypatch):
    """EEMFlow+'s encoder is EEMFlow's: with 5 input channels it runs on the encoder kernels (enc1 / Winograd / enc2);
    EEM_PLUS_GENERIC_ENC=1 (read when the weights are loaded) keeps it on the generic conv.  Same coarsest flow to fp32
    round-off; the finer levels may differ where a warp mask sits on its >= 1.0 threshold."""
    h, w = 256, 320
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(61, 1, h, w, bins=5))

    def run():
        net = make_net(62, 5)
        net.change_imagesize((h, w))
        with torch.no_grad():
            return torch.stack(net(e1, e2)[1]).cpu().numpy()
    fast = run()
    monkeypatch.setenv("EEM_PLUS_GENERIC_ENC", "1")
    gen = run()
    err = np.abs(fast - gen)
    assert err[0].max() < 1e-4
    for i in range(1, 5):
        assert np.median(err[i]) < 2e-3 and (err[i] > 0.25).mean() < 0.02 and err[i].max() < 2.0, i   # bounds of the golden test


def test_grouped_decoder_launch_equals_one_launch_per_group(monkeypatch):
    """The decoder's three 32 -> 32 groups of a layer (+ channel shuffle, EEMFlow+.py:52-63) run as ONE launch of the LDS-tiled kernel
    (blockIdx.y = (group, cout tile)); EEM_PLUS_NO_GROUPED=1 (read per forward) launches them one by one.  Same kernel, same k order
    per output: the same flows."""
    h, w = 256, 320
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(71, 1, h, w, bins=5))
    net = make_net(72, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        grouped = torch.stack(net(e1, e2)[1]).cpu().numpy()
        monkeypatch.setenv("EEM_PLUS_NO_GROUPED", "1")
        apart = torch.stack(net(e1, e2)[1]).cpu().numpy()
    assert np.abs(grouped - apart).max() < 1e-6 and np.abs(apart).max() > 1e-3


def test_fused_warp_blend_copy_equals_the_three_launches(monkeypatch):
    """cdc_model's torch_warp + mask blend (cdc_utils.py:163-173) and the copy of flow_up into the decoder's input are one launch
    (warp_blend_kernel); EEM_PLUS_NO_FUSE=1 (read per forward) keeps warp_kernel / blend_kernel / copy_channels_kernel.  The same
    operations in the same order per pixel, and the five full-resolution predictions leave through one vectorised launch whose
    per-output arithmetic is upflow_kernel's: bitwise the same flows."""
    h, w = 256, 320
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(81, 2, h, w, bins=5))
    net = make_net(82, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        fused = torch.stack(net(e1, e2)[1]).cpu().numpy()
        monkeypatch.setenv("EEM_PLUS_NO_FUSE", "1")
        apart = torch.stack(net(e1, e2)[1]).cpu().numpy()
    assert np.array_equal(fused, apart) and np.abs(apart).max() > 1e-3


def test_wide_dense_estimator_convs_on_the_small_grid_kernel(monkeypatch):
    """The dense estimator's 128 / 160 / 176 / 184-channel convs (cdc_utils.py) on the coarse pyramid levels run on the small-grid
    kernel (46 fragments per wave; EEM_PLUS_TAIL_MAXCIN=100, read per call, keeps them on the LDS-tiled / few-cout kernels): the same
    flows to summation order, level by level teacher-forced from the same flow_init (the chained forward amplifies the reference's own
    `>= 1.0` mask discontinuity)."""
    h, w = 256, 320
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(85, 2, h, w, bins=5))
    net = make_net(86, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        net(e1, e2)
        fi = {l: net.stage(f"flow_init{l}").clone() for l in (5, 4, 3)}
        new = {l: [t.clone() for t in net.level(l, fi[l])] for l in (5, 4, 3)}
        monkeypatch.setenv("EEM_PLUS_TAIL_MAXCIN", "100")
        old = {l: [t.clone() for t in net.level(l, fi[l])] for l in (5, 4, 3)}
    assert not torch.equal(new[5][1], old[5][1])              # (the switch did switch)
    for l in (5, 4, 3):
        for a, b in zip(new[l], old[l]):
            assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("b,h,w,small_maxpx", [(2, 256, 320, None), (1, 200, 300, None), (1, 256, 320, "0"), (1, 720, 1280, None)])
def test_winograd_kernel_of_the_fine_levels_equals_the_lds_tiled_kernels(monkeypatch, b, h, w, small_maxpx):
    """conv_wnc.hip - F(2x2) Winograd over input chunks of 32 channels, the decoder's groups / 32-cout slices as the jobs of one launch,
    the <= 16-cout layers on the 16x16x4 MFMA, the 176- and 184-channel inputs through an overlapping last chunk with zeroed weights -
    against the LDS-tiled / bf16-piece / few-cout kernels it replaces (EEM_NO_WNC=1), level by level teacher-forced from the same
    flow_init.  EEM_PLUS_WNC_MINPX=0 sends every level's layers through it (partial tiles at 4 x 5 .. 64 x 80 maps); both block tiles
    (EEM_WNC_SMALL_MAXPX=0: the 4 x 64 one everywhere); all three switches are read per call."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(95, b, h, w, bins=5))
    net = make_net(96, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        net(e1, e2)
        fi = {l: net.stage(f"flow_init{l}").clone() for l in (5, 4, 3, 2)}
        if (h, w) != (720, 1280):
            monkeypatch.setenv("EEM_PLUS_WNC_MINPX", "0")       # (at 1280x720 the default policy: levels 3 and 2)
        if small_maxpx is not None:
            monkeypatch.setenv("EEM_WNC_SMALL_MAXPX", small_maxpx)
        new = {l: [t.clone() for t in net.level(l, fi[l])] for l in (5, 4, 3, 2)}
        monkeypatch.setenv("EEM_NO_WNC", "1")
        old = {l: [t.clone() for t in net.level(l, fi[l])] for l in (5, 4, 3, 2)}
    assert not torch.equal(new[2][1], old[2][1])              # (the switch did switch)
    for l in (5, 4, 3, 2):
        for x, y in zip(new[l], old[l]):
            assert float((x - y).abs().max()) < 2e-4 * max(1.0, float(y.abs().max())), l


def test_level_units_on_the_side_stream_equal_the_chain(monkeypatch):
    """EEM_PLUS_SIDE=1 (read per forward; opt-in, measured slower): the 1x1 projections and rconv of every level run on a side stream
    beside the coarse levels' chain, in the buffers each level owns - the same launches, the same flows, bitwise."""
    h, w = 256, 320
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(97, 2, h, w, bins=5))
    net = make_net(98, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        chain = torch.stack(net(e1, e2)[1]).cpu().numpy()
        monkeypatch.setenv("EEM_PLUS_SIDE", "1")
        side = [torch.stack(net(e1, e2)[1]).cpu().numpy() for _ in range(3)]
    assert all(np.array_equal(chain, s_) for s_ in side) and np.abs(chain).max() > 1e-3


@pytest.mark.parametrize("h,w", [(256, 320), (200, 296)])
def test_one_launch_feature_levels_and_a_level6_fill_per_shape_equal_the_chain(monkeypatch, h, w):
    """Feature levels 4 .. 6 (three avg_pool2d, EEMFlow+.py:170-175) from ONE launch that reads level 3 once (EEM_POOL_CHAIN=1, read per
    call: three launches), and the zeros of level 6's two flow channels (:179) filled once per (batch, map size) instead of per forward -
    checked by changing the shape in between (the buffer then holds another layout's values where the zeros have to be) and coming back:
    the five predictions bit for bit."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(101, 2, h, w, bins=5))
    o1, o2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(102, 1, 384, 512, bins=5))
    net = make_net(103, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        first = torch.stack(net(e1, e2)[1]).cpu().numpy()
        again = torch.stack(net(e1, e2)[1]).cpu().numpy()                    # (no fill this time)
        net.change_imagesize((384, 512))
        other = torch.stack(net(o1, o2)[1]).cpu().numpy()
        net.change_imagesize((h, w))
        back = torch.stack(net(e1, e2)[1]).cpu().numpy()
        monkeypatch.setenv("EEM_POOL_CHAIN", "1")
        chain = torch.stack(net(e1, e2)[1]).cpu().numpy()
    assert np.array_equal(first, again) and np.array_equal(first, back) and np.array_equal(first, chain)
    assert np.abs(first).max() > 1e-3 and np.abs(other).max() > 1e-3


@pytest.mark.parametrize("n,h,w", [(3, 256, 320), (2, 100, 150), (1, 128, 192)])
def test_forward_many_equals_the_batched_forward(n, h, w):
    """EEMFlow_cdc.forward_many (eemplus_forward_many): n independent batch-1 samples in their own tensors through one batch-n chain -
    bitwise the five predictions of `forward` on the samples stacked into a batch (padded sizes included: 100 x 150 pads to 128 x 192);
    argument errors for an empty list, mixed shapes and batched frames."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(87, n, h, w, bins=5))
    net = make_net(88, 5)
    net.change_imagesize((h, w))
    with torch.no_grad():
        batched = torch.stack(net(e1, e2)[1]).clone()                       # [5, n, 2, h, w]
        frames = [(e1[i:i + 1].clone(), e2[i:i + 1].clone()) for i in range(n)]
        many = net.forward_many(frames)
    assert len(many) == n
    for i, ((a, b), preds) in enumerate(many):
        assert a is frames[i][0] and len(preds) == 5 and preds[0].shape == (1, 2, h, w)
        assert torch.equal(torch.stack(preds)[:, 0], batched[:, i])
    assert float(batched.abs().max()) > 1e-3
    with pytest.raises(ValueError):
        net.forward_many([])
    with pytest.raises(ValueError):
        net.forward_many([(e1, e2)] if n > 1 else [(torch.cat([e1, e1]), torch.cat([e2, e2]))])
