"""Parity of the HIP path (through the C ABI) against the oracle and the reference-generated golden
vectors.  Needs a real MI355X: run with `pytest -m gpu`.

Tolerances (SURVEY.md 8c / BASELINE.json north_star): flow max-abs <= 1e-3 (we assert 1e-4, the path
is exact fp32), intermediate features <= 1e-4, integer event-bin indices bit-exact."""
import ctypes

import numpy as np
import pytest
import torch

from eemflow_amd import EEMFlow, EventSequence, EventSequenceToVoxelGrid_Pytorch, _lib
from eemflow_amd.weights import seeded_state_dict, synthetic_voxel_pair
from oracle import eemflow_oracle as O

pytestmark = pytest.mark.gpu
FLOW_TOL = 1e-4      # north star allows 1e-3
FEAT_TOL = 1e-4
DEV = "cuda:0"


def make_net(seed, graph=True):
    sd = seeded_state_dict(seed)
    net = EEMFlow("", groups=5, n_first_channels=5).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.use_graph = graph
    return net.to(DEV), sd


def maxerr(a, b):
    return float((torch.as_tensor(a).cpu().float() - torch.as_tensor(b).cpu().float()).abs().max())


def test_library_is_the_loaded_code():
    _lib.lib()
    assert any("libeemflow_hip.so" in ln for ln in open("/proc/self/maps"))


@pytest.mark.parametrize("tag", ["128x192", "260x346", "100x150"])
def test_forward_vs_golden(golden, tag):
    g = golden(f"eemflow_fwd_{tag}.npz")
    h, w = g["hw"].tolist()
    net, _ = make_net(int(g["seed"]))
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w))
    with torch.no_grad():
        (r1, r2), preds = net(e1, e2)
    assert r1 is e1 and r2 is e2 and len(preds) == 1
    flow = preds[0]
    assert flow.shape == (int(g["batch"]), 2, h, w)
    b = int(g["batch"])
    if "f11" in g.files:        # stage tensors of the first event volume
        for name in ("f11", "f12", "f13"):
            assert maxerr(net.stage(name)[:b], g[name]) < FEAT_TOL, name
        assert maxerr(net.stage("f13")[b:], g["f23"]) < FEAT_TOL
    for k in (1, 2, 3):
        pool = net.stage(f"pool_{k}")
        assert maxerr(pool[:b], g[f"pool1_{k}"]) < FEAT_TOL
        assert maxerr(pool[b:], g[f"pool2_{k}"]) < FEAT_TOL
        cat = net.stage(f"cat_{k}")
        assert maxerr(cat[:, :53], g[f"cv_{k}"]) < FEAT_TOL
        assert maxerr(cat[:, 53:], g[f"r_{k}"]) < FEAT_TOL
        assert maxerr(net.stage("flowcat")[:, 2 * (k - 1):2 * k], g[f"flow_{k}"]) < FEAT_TOL
    assert maxerr(net.stage("coarse"), g["coarse"]) < FEAT_TOL
    assert maxerr(flow, g["flow"]) < FLOW_TOL


@pytest.mark.parametrize("b,h,w", [(1, 64, 64), (3, 72, 200), (2, 130, 70), (1, 200, 333)])
def test_forward_vs_oracle_ragged(b, h, w):
    net, sd = make_net(11)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(12, b, h, w))
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd), e1, e2)
    assert maxerr(flow, ref) < FLOW_TOL


@pytest.mark.parametrize("frames_in_flight", [1, 4])
def test_forward_full_size_1280x720(frames_in_flight):
    """BASELINE config[1]: 1x5x720x1280, checked directly against the oracle (a few seconds of CPU) - in the single-frame launch
    configuration (frames_in_flight = 1, the default) AND in the one bench.py times (frames_in_flight = 4: fewer, longer encoder
    blocks and the throughput kernels on every stride-1 layer)."""
    h, w = 720, 1280
    net, sd = make_net(21)
    net.frames_in_flight = frames_in_flight
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(22, 1, h, w))
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0]
        ref, st = O.eemflow_forward(O.to_torch_sd(sd), e1, e2, keep=True)
    for name in ("f11", "f12", "f13"):
        assert maxerr(net.stage(name)[:1], st[name]) < FEAT_TOL, name
    assert maxerr(flow, ref) < FLOW_TOL
    # EPE against a smooth synthetic ground truth agrees with the oracle's (north star: within 1e-3)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    gt = np.stack([3 * np.sin(2 * np.pi * xx / w), 3 * np.cos(2 * np.pi * yy / h)])
    epe_hip = O.flow_error_dense(gt, flow[0].cpu().numpy())[0]
    epe_ref = O.flow_error_dense(gt, ref[0].numpy())[0]
    assert abs(epe_hip - epe_ref) < 1e-3


def test_padder_mismatch_quirk():
    """change_imagesize for one size, forward on another (train_mvsec.py:230 does this)."""
    net, sd = make_net(31)
    net.change_imagesize((260, 346))             # pad [19,19,0,60]
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(32, 1, 196, 282))   # padded 256x320
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd), e1, e2, image_size=(260, 346))
    assert maxerr(flow, ref) < FLOW_TOL


def test_graph_equals_eager_and_is_repeatable():
    h, w = 128, 192
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(42, 2, h, w))
    outs = []
    for graph in (True, False):
        net, _ = make_net(41, graph=graph)
        net.change_imagesize((h, w))
        with torch.no_grad():
            a = net(e1, e2)[1][0].clone()
            b = net(e1, e2)[1][0].clone()
        assert torch.equal(a, b)                 # no atomics on this path: bitwise repeatable
        outs.append(a)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("mask", ["0", "7", None])
def test_frames_in_flight_hint_changes_grids_not_results(monkeypatch, mask):
    """eemflow_set_frames_in_flight >= 3 launches the persistent encoder kernels on fewer blocks (more tiles each) and, by default, moves
    the C = 32 / 64 stride-1 layers from Winograd F(2x2,3x3) to F(4x4,3x3) (fewer, longer blocks: less CU time, more latency).  With
    the form pinned (EEM_WINO4_LAYERS) it is the same tiles and the same arithmetic: the flow is bitwise the same, at the headline
    size and at a ragged one; with the default policy the two forms agree to fp32 round-off and each setting is bitwise stable."""
    if mask is None:
        monkeypatch.delenv("EEM_WINO4_LAYERS", raising=False)
    else:
        monkeypatch.setenv("EEM_WINO4_LAYERS", mask)             # read when the weights are loaded
    for (h, w, seed) in ((720, 1280, 51), (260, 346, 52)):
        e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(seed, 1, h, w))
        net, _ = make_net(43)
        net.change_imagesize((h, w))
        outs = []
        for n in (1, 4, 1):
            net.frames_in_flight = n
            with torch.no_grad():
                outs.append(net(e1, e2)[1][0].clone())
                outs.append(net(e1, e2)[1][0].clone())        # replay of the graph captured for this setting
        if mask is None:
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3]) and torch.equal(outs[0], outs[4]) and torch.equal(outs[0], outs[5])
            assert maxerr(outs[0], outs[2]) < 2e-5
        else:
            assert all(torch.equal(outs[0], o) for o in outs[1:])
        assert graph_stats(net)[0] == 3                       # the hint flipped twice: the cached graph was dropped each time


@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (2, 260, 346), (1, 100, 150)])
def test_light_block_stride2_kernel_equals_the_chunked_one(monkeypatch, b, h, w):
    """conv_s2.hip (register-resident weights, the whole tile by one DMA wait) against conv_enc2.hip's chunked kernel
    (EEM_NO_S2W=1, read per launch): same k order per output - bitwise the same stage tensors and flow."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(55, b, h, w))
    outs = []
    monkeypatch.setenv("EEM_S2W_64", "1")                    # pconv3_1 too (it stays on the chunked kernel by default)
    monkeypatch.setenv("EEM_NO_BX3", "1")                    # (pconv2_1's default is the bf16-piece kernel, conv_bx3.hip)
    for off in ("0", "1"):
        monkeypatch.setenv("EEM_NO_S2W", off)
        net, _ = make_net(44, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
        outs.append((flow, net.stage("f12").clone(), net.stage("f13").clone()))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (2, 260, 346), (1, 100, 150)])
def test_bf16_piece_stride2_kernel_equals_the_fp32_one(monkeypatch, b, h, w):
    """conv_bx3.hip runs pconv2_1 (EEMFlow.py:77) on the bf16 matrix pipe: every fp32 operand as three bf16 pieces (exact), six of the
    nine piece products (the dropped ones are below 2^-24 of a product), fp32 accumulation - fp32 arithmetic in another order, not a
    reduced precision: against the fp32-MFMA kernel (EEM_NO_BX3=1, read per launch) the layer's output agrees to a few ulp of its
    largest sums, and the flow far inside the 1e-4 budget."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(57, b, h, w))
    outs = []
    for off in ("0", "1"):
        monkeypatch.setenv("EEM_NO_BX3", off)                            # both stride-2 layers: pconv2_1 and pconv3_1 (EEMFlow.py:77,80)
        net, _ = make_net(46, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
        outs.append((flow, net.stage("a2").clone(), net.stage("f12").clone(), net.stage("a3").clone()))
    (flow_b, a2_b, f12_b, a3_b), (flow_f, a2_f, f12_f, a3_f) = outs
    assert not torch.equal(a2_b, a2_f) and not torch.equal(a3_b, a3_f)   # (the switch did switch)
    scale = float(a2_f.abs().max())
    assert scale > 1e-3 and maxerr(a2_b, a2_f) < 2e-6 * max(scale, 1.0)  # K = 144 sums of either kernel: ~2e-7 of sum |a b|
    assert maxerr(f12_b, f12_f) < 1e-5 * max(float(f12_f.abs().max()), 1.0)
    assert maxerr(a3_b, a3_f) < 1e-5 * max(float(a3_f.abs().max()), 1.0)
    assert maxerr(flow_b, flow_f) < 2e-5


@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (3, 512, 768), (2, 200, 256)])
def test_decoder_wide_convs_on_the_winograd_kernel_equal_the_small_grid_ones(monkeypatch, b, h, w):
    """The decoders' conv1 69 -> 100 and conv5 100 -> 64 (EEMFlow.py:38-71) run on the Winograd F(2x2) kernel of conv_wnc.hip where the
    1/64 grid's rows are 16-byte multiples - all three decoders' 32-cout slices as the jobs of one launch, the streams packed from the
    device-resident weights (ensure_dec_wnc) from four samples per launch on; EEM_DEC_WNC=1 / 0 (read per call): at every batch / never.  The same convolution in
    another arithmetic: the decoder outputs agree to summation-order round-off, the flow far inside the 1e-4 budget; and against the
    oracle.  After new weights arrive on the device (load_state_dict) the streams are re-packed: same agreement with the other seed."""
    from oracle import eemflow_oracle as O
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(61, b, h, w))
    outs = {}
    for off in ("1", "0"):
        monkeypatch.setenv("EEM_DEC_WNC", off)
        net, sd = make_net(62, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
            fc = net.stage("flowcat").clone()
            sd2 = seeded_state_dict(63)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
            flow2 = net(e1, e2)[1][0].clone()
        outs[off] = (flow, fc, flow2)
    (f_w, fc_w, f2_w), (f_s, fc_s, f2_s) = outs["1"], outs["0"]
    assert not torch.equal(fc_w, fc_s)                                   # (the switch did switch)
    scale = max(float(fc_s.abs().max()), 1.0)
    assert maxerr(fc_w, fc_s) < 2e-5 * scale and maxerr(f_w, f_s) < 2e-5 and maxerr(f2_w, f2_s) < 2e-5
    assert not torch.equal(f_w, f2_w)
    if h * w <= 512 * 768:
        ref = O.eemflow_forward(O.to_torch_sd(sd2), e1.cpu(), e2.cpu(), image_size=(h, w))[0]
        assert maxerr(f2_w, ref) < FLOW_TOL


def test_bf16_piece_kernel_and_an_infinity_in_the_input(monkeypatch):
    """ADVICE round 4 item 5 / VERDICT round 5: the three-piece split of conv_bx3.hip computes a - (a & 0xffff0000) - for a = inf that is
    inf - inf = NaN, so an infinite operand leaves the kernel as NaN where the fp32 MFMA kernel leaves an infinity.  The documented
    contract (conv_bx3.hip header): a non-finite input gives a NON-FINITE output at exactly the positions where the fp32 kernel's output
    is non-finite, and every other value is unchanged - nothing non-finite is ever turned into a number (the training step's
    inf / NaN skip, train_mvsec.py:253-258 GradScaler semantics, sees it either way)."""
    h, w = 128, 192
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(59, 1, h, w))
    e1 = e1.clone()
    e1[0, 2, 40, 100] = float("inf")
    e2 = e2.clone()
    e2[0, 1, 90, 30] = float("-inf")
    outs = []
    for off in ("0", "1"):
        monkeypatch.setenv("EEM_NO_BX3", off)
        net, _ = make_net(47, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            net(e1, e2)
        outs.append((net.stage("f11").clone(), net.stage("a2").clone(), net.stage("a3").clone()))
    (f11_b, a2_b, a3_b), (f11_f, a2_f, a3_f) = outs
    assert not torch.isfinite(f11_f).all()                       # the infinity reached the layer's input
    for got, ref in ((a2_b, a2_f), (a3_b, a3_f)):
        bad_g, bad_r = ~torch.isfinite(got), ~torch.isfinite(ref)
        assert bad_r.any() and torch.equal(bad_g, bad_r)
        ok = ~bad_r
        assert maxerr(got[ok], ref[ok]) < 1e-5 * max(1.0, float(ref[ok].abs().max()))


@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (2, 260, 346), (1, 100, 150)])
def test_bf16_piece_stride1_kernels_equal_the_winograd_ones(monkeypatch, b, h, w):
    """pconv2_2 (32 -> 32) and pconv3_2 (64 -> 64, EEMFlow.py:78,80) as direct convolutions on the bf16 matrix pipe (conv_bx3.hip:
    weight fragments through an LDS ring; at 64 channels two groups of four waves split the channels) against the F(4x4,3x3) /
    F(2x2,3x3) fp32 kernels (EEM_BX3_S1=0, read per launch; a mask: 1 = the 32-channel layer, 2 = the 64-channel one)."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(58, b, h, w))
    outs = []
    for mask in ("3", "0"):
        monkeypatch.setenv("EEM_BX3_S1", mask)
        net, _ = make_net(47, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
        outs.append((flow, net.stage("b2").clone(), net.stage("b3").clone(), net.stage("f13").clone()))
    (flow_b, b2_b, b3_b, f13_b), (flow_w, b2_w, b3_w, f13_w) = outs
    assert not torch.equal(b2_b, b2_w) and not torch.equal(b3_b, b3_w)   # (the switch did switch)
    for got, ref in ((b2_b, b2_w), (b3_b, b3_w), (f13_b, f13_w)):        # (Winograd's own error is the larger one)
        scale = float(ref.abs().max())
        assert scale > 1e-3 and maxerr(got, ref) < 2e-5 * max(scale, 1.0)
    assert maxerr(flow_b, flow_w) < 2e-5


def graph_stats(net):
    gs = (ctypes.c_longlong * 3)()
    _lib.check(_lib.lib().eemflow_graph_stats(net._ctx, ctypes.byref(gs)))
    return list(gs)                              # captures, replays, io-table rewrites


def test_graph_cache_survives_fresh_buffers_and_keeps_several_shapes():
    """The evaluation loop of test_mvsec.py:580-597 brings new tensors every sample: the cached graph is keyed on shapes, the caller's
    buffers reach it through the context's io table.  Alternating sizes keep one graph each; a larger size that moves the workspace
    drops them; a training forward in between does not."""
    net, sd = make_net(45)
    tsd = O.to_torch_sd(sd)
    shapes = [(2, 128, 192), (1, 96, 128)]          # the larger one first: the second fits the workspace, nothing moves
    refs, inputs = {}, {}
    for b, h, w in shapes:
        e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(46 + h, b, h, w))
        with torch.no_grad():
            refs[(b, h, w)] = O.eemflow_forward(tsd, e1, e2)[0]
        inputs[(b, h, w)] = (e1, e2)
    hold = []
    with torch.no_grad():
        for rep in range(4):
            for b, h, w in shapes:
                net.change_imagesize((h, w))
                e1, e2 = (t.to(DEV).clone() for t in inputs[(b, h, w)])        # fresh device tensors every call
                flow = net(e1, e2)[1][0]
                hold += [e1, e2, flow]                                          # keep them alive: the allocator must hand out new blocks
                assert maxerr(flow, refs[(b, h, w)]) < FLOW_TOL
    cap, rep, io = graph_stats(net)
    assert cap == 2 and rep == 8 and io == 8, (cap, rep, io)
    # same buffers again: no table rewrite
    with torch.no_grad():
        net.change_imagesize((96, 128))
        e1, e2 = hold[-3], hold[-2]
        out_a = net(e1, e2)[1][0].clone()
    net.train()
    net(e1, e2)[1][0].sum().backward()           # autograd forward + backward on the same context: graphs stay
    net.eval()
    with torch.no_grad():
        out_b = net(e1, e2)[1][0]
    assert torch.equal(out_a, out_b) and graph_stats(net)[0] == 2
    # a larger frame moves the workspace: every cached graph is dropped and rebuilt on demand
    b, h, w = 1, 256, 320
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(49, b, h, w))
    with torch.no_grad():
        net.change_imagesize((h, w))
        big = net(e1.to(DEV), e2.to(DEV))[1][0]
        assert maxerr(big, O.eemflow_forward(tsd, e1, e2)[0]) < FLOW_TOL
        net.change_imagesize((128, 192))
        e1, e2 = (t.to(DEV) for t in inputs[(2, 128, 192)])
        assert maxerr(net(e1, e2)[1][0], refs[(2, 128, 192)]) < FLOW_TOL
    assert graph_stats(net)[0] == 4


def test_weights_reload_after_update():
    h, w = 64, 128
    net, sd = make_net(51)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(52, 1, h, w))
    with torch.no_grad():
        f0 = net(e1.to(DEV), e2.to(DEV))[1][0].clone()
        sd2 = seeded_state_dict(53)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        f1 = net(e1.to(DEV), e2.to(DEV))[1][0]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd2), e1, e2)
    assert maxerr(f1, ref) < FLOW_TOL and maxerr(f0, f1) > 1e-3


def test_out_mesh_size_training_shape():
    net, sd = make_net(61)
    net.out_mesh_size = True
    net.change_imagesize((128, 128))
    for p in net.parameters():
        p.requires_grad_(False)
    net.train()
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(62, 1, 128, 128))
    with torch.no_grad():
        flow = net(e1.to(DEV), e2.to(DEV))[1][0]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd), e1, e2, out_size=(16, 16))
    assert flow.shape == (1, 2, 16, 16) and maxerr(flow, ref) < FLOW_TOL


def test_decoder_golden(golden):
    g = golden("decoder.npz")
    net, _ = make_net(int(g["seed"]))
    net.change_imagesize((64, 64))
    ctx = net._context(torch.device(DEV))
    x = torch.from_numpy(g["x"]).to(DEV)
    out = torch.empty(x.shape[0], 2, x.shape[2], x.shape[3], device=DEV)
    _lib.check(_lib.lib().eemflow_decoder(ctx, 2, x.data_ptr(), x.shape[0], x.shape[2], x.shape[3], out.data_ptr(),
                                          _lib.current_stream_ptr(torch.device(DEV))))
    assert maxerr(out, g["y"]) < FEAT_TOL


def test_local_corr_golden(golden):
    g = golden("local_corr.npz")
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    b, c, h, w = x.shape
    out = torch.empty(b, 53, h, w, device=DEV)
    _lib.check(_lib.lib().eemflow_local_corr53(x.data_ptr(), y.data_ptr(), b, c, h, w, out.data_ptr(),
                                               _lib.current_stream_ptr(torch.device(DEV))))
    assert maxerr(out, g["cv53"]) < 1e-5


@pytest.mark.parametrize("b,c,h,w", [(1, 32, 192, 320), (2, 20, 181, 203)])
def test_local_corr_large_maps_vs_oracle(b, c, h, w):
    """Maps of >= 32768 pixels take the tiled correlation kernel (16x16 tiles, f2 patch in LDS); ragged tiles, a channel count
    that is not a multiple of the 8-channel chunk, batch 2 - against the oracle."""
    gen = torch.Generator().manual_seed(7)
    x, y = torch.randn(b, c, h, w, generator=gen), torch.randn(b, c, h, w, generator=gen)
    out = torch.empty(b, 53, h, w, device=DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    _lib.check(_lib.lib().eemflow_local_corr53(xd.data_ptr(), yd.data_ptr(), b, c, h, w, out.data_ptr(),
                                               _lib.current_stream_ptr(torch.device(DEV))))
    assert maxerr(out, O.local_corr53(x, y)) < 1e-5


@pytest.mark.parametrize("shape,size", [((2, 2, 12, 20), (720, 1280)), ((1, 2, 5, 6), (260, 346)),
                                        ((3, 2, 4, 7), (16, 16)), ((1, 2, 9, 9), (5, 31))])
def test_upsample_vs_torch(shape, size):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    net, _ = make_net(1)
    got = net.upsample_flow(x.to(DEV), size)
    ref = O.upsample_flow(x, size)
    assert maxerr(got, ref) < 1e-5


VOX_CASES = ["n20k", "n20k_pol01", "n1", "n2_dt0", "n3_dt0", "n500_raw", "n300_bins3", "n400_unsorted", "n64_const"]


@pytest.mark.parametrize("name", VOX_CASES)
def test_voxelizer_golden(golden, name):
    g = golden("voxel.npz")
    h, w, bins, norm = g[f"{name}_hwb"].tolist()
    seq = EventSequence(None, {"height": h, "width": w}, features=g[f"{name}_events"].copy(),
                        timestamp_multiplier=1e6, convert_to_relative=True)
    vox = EventSequenceToVoxelGrid_Pytorch(num_bins=bins, normalize=bool(norm), gpu=True, forkserver=False)
    grid, il, ir = vox(seq, return_indices=True)
    il, ir = il.cpu().numpy(), ir.cpu().numpy()
    assert np.array_equal(il[il >= 0], g[f"{name}_idx_left"])      # int64 indices: bit-exact
    assert np.array_equal(ir[ir >= 0], g[f"{name}_idx_right"])
    ref = g[f"{name}_grid"]
    got = grid.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), atol=2e-5, rtol=1e-5)
    assert np.array_equal(got != 0, ref != 0) or not norm       # same support


def test_voxelizer_full_size_properties():
    """1280x720, N=2e6: indices against the oracle bit-exact; raw grid sums to the vote sum;
    normalised grid has zero mean / unit unbiased std over its non-zero voxels."""
    rng = np.random.default_rng(0)
    n, h, w = 2_000_000, 720, 1280
    ev = np.stack([np.sort(rng.uniform(0, 0.05, n)), rng.integers(0, w, n).astype(np.float64),
                   rng.integers(0, h, n).astype(np.float64), rng.integers(0, 2, n) * 2.0 - 1.0], 1)
    seq = EventSequence(None, {"height": h, "width": w}, features=ev.copy(), timestamp_multiplier=1e6,
                        convert_to_relative=True)
    feats = O.event_sequence(ev, 1e6, True)
    assert np.array_equal(seq.features, feats)
    il_ref, vl, ir_ref, vr = O.voxel_indices(feats, 5, h, w)
    raw, il, ir = EventSequenceToVoxelGrid_Pytorch(5, gpu=True, normalize=False, forkserver=False)(seq, True)
    il, ir = il.cpu().numpy(), ir.cpu().numpy()
    assert np.array_equal(il[il >= 0], il_ref) and np.array_equal(ir[ir >= 0], ir_ref)
    total = float(vl.astype(np.float64).sum() + vr.astype(np.float64).sum())
    assert abs(float(raw.double().sum()) - total) < 1e-2
    ref = O.voxelize(feats, 5, h, w, normalize=False)
    np.testing.assert_allclose(raw.cpu().numpy(), ref, atol=1e-5)
    norm = EventSequenceToVoxelGrid_Pytorch(5, gpu=True, normalize=True, forkserver=False)(seq)
    nz = norm[norm != 0].double()
    assert abs(float(nz.mean())) < 1e-4 and abs(float(nz.std()) - 1.0) < 1e-4


def _clustered_events(seed, n, h, w, wrap=False):
    """Time-sorted events, 70% of them inside one 40x40 patch (one hot band), optionally some with x in [w, w+4)
    on rows above the last one (the reference's flat index then lands in the next row)."""
    rng = np.random.default_rng(seed)
    hot = rng.random(n) < 0.7
    ph, pw = min(40, h), min(40, w)
    x = np.where(hot, rng.integers(0, pw, n) + (w - pw) // 2, rng.integers(0, w, n))
    y = np.where(hot, rng.integers(0, ph, n) + (h - ph) // 2, rng.integers(0, h, n))
    if wrap:
        k = rng.random(n) < 0.05
        x = np.where(k, w + rng.integers(0, 4, n), x)
        y = np.where(k, np.minimum(y, h - 2), y)
    t = np.sort(rng.uniform(0, 0.05, n))
    return np.stack([t, x.astype(np.float64), y.astype(np.float64), rng.integers(0, 2, n) * 2.0 - 1.0], 1)


@pytest.mark.parametrize("h,w,bins,n,wrap", [(720, 1280, 5, 300_000, False), (480, 640, 15, 200_000, False),
                                             (260, 346, 5, 50_000, True), (37, 53, 3, 5_000, True),
                                             (1100, 1300, 20, 100_000, False), (1100, 1300, 31, 50_000, False)])
def test_voxelizer_binned_path_equals_direct_path_and_oracle(monkeypatch, h, w, bins, n, wrap):
    """The LDS-band voxelizer (count / bin / band kernels) against the direct atomic kernel of the same library
    (EEM_VOX_DIRECT=1) and the oracle, on clustered events: identical int64 indices, same support, values to fp32
    summation-order round-off, raw and normalised.  The last two cases: bands of 112 KiB of LDS; a grid too large for bands (both runs take the direct kernel)."""
    ev = _clustered_events(h * 7 + bins, n, h, w, wrap)
    seq = EventSequence(None, {"height": h, "width": w}, features=ev.copy(), timestamp_multiplier=1e6,
                        convert_to_relative=True)
    out = {}
    for mode in ("binned", "binned_two_band_passes", "binned_norm_pass", "direct"):
        if mode == "direct":
            monkeypatch.setenv("EEM_VOX_DIRECT", "1")
        # normalisation: by default two band passes for sparse volumes, else a pass over the grid; both forced here on every case
        monkeypatch.setenv("EEM_VOX_TWOPASS", {"binned_two_band_passes": "1", "binned_norm_pass": "0"}.get(mode, "8"))
        raw, il, ir = EventSequenceToVoxelGrid_Pytorch(bins, gpu=True, normalize=False, forkserver=False)(seq, True)
        norm = EventSequenceToVoxelGrid_Pytorch(bins, gpu=True, normalize=True, forkserver=False)(seq)
        out[mode] = (raw.cpu().numpy(), il.cpu().numpy(), ir.cpu().numpy(), norm.cpu().numpy())
    b, d = out["binned"], out["direct"]
    for other in ("binned_two_band_passes", "binned_norm_pass"):
        np.testing.assert_allclose(out[other][3], b[3], atol=3e-5, rtol=1e-5)
    assert np.array_equal(b[1], d[1]) and np.array_equal(b[2], d[2])
    il_ref, _, ir_ref, _ = O.voxel_indices(seq.features, bins, h, w)
    assert np.array_equal(b[1][b[1] >= 0], il_ref) and np.array_equal(b[2][b[2] >= 0], ir_ref)
    ref_raw = O.voxelize(seq.features, bins, h, w, normalize=False)
    ref_norm = O.voxelize(seq.features, bins, h, w, normalize=True)
    scale = max(1.0, float(np.abs(ref_raw).max()))
    for got in (b[0], d[0]):
        np.testing.assert_allclose(got, ref_raw, atol=2e-6 * scale * 8, rtol=0)
    for got in (b[3], d[3]):
        np.testing.assert_allclose(got, ref_norm, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("direct", [False, True])
def test_voxelizer_out_of_image_events_keep_flat_index_semantics_and_never_write_outside(monkeypatch, direct):
    """The reference adds votes at the FLAT index x + y*w + bin*h*w of the flattened grid (loader_utils.py:505-521): an event with
    y >= h (or y < 0) lands in a neighbouring bin's plane while that index is inside the grid, and index_add_ raises once it is not.
    Both kernels reproduce the in-grid part and drop the rest; the grid sits between two guard regions that must stay untouched."""
    from eemflow_amd.voxelizer import voxelize_device
    h, w, bins, n = 60, 80, 5, 20_000
    rng = np.random.default_rng(3)
    t = np.sort(rng.uniform(0, 1000.0, n))
    t[0] = 0.0
    x = rng.integers(0, w, n).astype(np.float64)
    y = rng.integers(0, h, n).astype(np.float64)
    bad = rng.random(n) < 0.1
    y = np.where(bad, np.where(rng.random(n) < 0.5, h + rng.integers(0, 150, n), -1.0 - rng.integers(0, 150, n)), y)
    ev = np.stack([t, x, y, rng.integers(0, 2, n) * 2.0 - 1.0], 1)
    if direct:
        monkeypatch.setenv("EEM_VOX_DIRECT", "1")
    plane, total = h * w, bins * h * w
    buf = torch.full((3 * total,), 7.0, device=DEV)
    grid = buf[total:2 * total].view(bins, h, w)
    voxelize_device(torch.from_numpy(ev).to(DEV), bins, h, w, normalize=False, out=grid)
    torch.cuda.synchronize()
    assert bool((buf[:total] == 7.0).all()) and bool((buf[2 * total:] == 7.0).all())
    ts = (bins - 1) * (t - t[0]) / (t[-1] - t[0])
    tl = np.floor(ts)
    d = (ts - tl).astype(np.float32)
    il = (x + y * w + tl * plane).astype(np.int64)
    ir = il + plane
    okl = (tl < bins) & (tl >= 0) & (il >= 0) & (il < total)
    okr = (tl + 1 < bins) & (tl >= 0) & (ir >= 0) & (ir < total)
    assert int((okl & bad).sum()) > 100 and int((~okl & bad).sum()) > 100          # both kinds of stray events are present
    ref = np.zeros(total, dtype=np.float64)
    np.add.at(ref, il[okl], (ev[okl, 3] * (np.float32(1.0) - d[okl])).astype(np.float64))
    np.add.at(ref, ir[okr], (ev[okr, 3] * d[okr]).astype(np.float64))
    np.testing.assert_allclose(grid.cpu().numpy().ravel(), ref, atol=1e-4)


def test_voxelizer_more_than_512_blocks(monkeypatch):
    """5.2e6 events: the binning kernel runs 8 events per thread and more than 512 blocks (run table sized by event count); the
    band kernel walks the slabs in groups of 1024.  Against the direct atomic kernel of the same library."""
    n, h, w, bins = 5_200_000, 720, 1280, 5
    rng = np.random.default_rng(11)
    ev = np.stack([np.sort(rng.uniform(0, 0.05, n)), rng.integers(0, w, n).astype(np.float64),
                   rng.integers(0, h, n).astype(np.float64), rng.integers(0, 2, n) * 2.0 - 1.0], 1)
    seq = EventSequence(None, {"height": h, "width": w}, features=ev, timestamp_multiplier=1e6, convert_to_relative=True)
    vox = EventSequenceToVoxelGrid_Pytorch(bins, gpu=True, normalize=True, forkserver=False)
    binned, il, ir = vox(seq, return_indices=True)
    monkeypatch.setenv("EEM_VOX_DIRECT", "1")
    direct, il2, ir2 = vox(seq, return_indices=True)
    assert torch.equal(il, il2) and torch.equal(ir, ir2)
    assert float((binned - direct).abs().max()) < 2e-4 and float(direct.abs().max()) > 1.0


@pytest.mark.parametrize("nstreams", [2, 6])
def test_voxelizer_alternating_streams_share_the_scratch_safely(nstreams):
    """Calls from one thread on several streams: up to four streams keep a scratch arena each, further ones take over the least
    recently used arena behind an event."""
    h, w, bins = 260, 346, 5
    seqs = [EventSequence(None, {"height": h, "width": w}, features=_clustered_events(90 + k, 40_000 + 7000 * k, h, w),
                          timestamp_multiplier=1e6, convert_to_relative=True) for k in range(4)]
    refs = [O.voxelize(s.features, bins, h, w, normalize=True) for s in seqs]
    vox = EventSequenceToVoxelGrid_Pytorch(bins, gpu=True, normalize=True, forkserver=False)
    streams = [torch.cuda.Stream(DEV) for _ in range(nstreams)]
    outs = []
    for rep in range(3):
        for k, s in enumerate(seqs):
            with torch.cuda.stream(streams[(rep * len(seqs) + k) % nstreams]):
                outs.append((k, vox(s)))
    torch.cuda.synchronize(DEV)
    for k, g in outs:
        np.testing.assert_allclose(g.cpu().numpy(), refs[k], atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("h,w,n1,n2", [(260, 346, 40_000, 61_000), (720, 1280, 200_000, 1_300_000), (720, 1280, 2_000_000, 3), (64, 64, 1, 500)])
def test_voxelizer_pair_equals_two_single_calls(monkeypatch, h, w, n1, n2):
    """eemflow_voxelize_pair (both volumes of a sample, one launch sequence; loader/HREM.py:226-232) against two eemflow_voxelize
    calls (normalised and raw): the same grids up to the order of the float adds inside a band (LDS atomics: not bitwise repeatable
    between two single calls either), also when the direct atomic kernel is forced (EEM_VOX_DIRECT=1: the pair then runs as two
    sequences), and against the oracle."""
    from eemflow_amd.voxelizer import voxelize_device, voxelize_pair_device
    bins = 5
    ev = [torch.from_numpy(np.ascontiguousarray(EventSequence(None, {"height": h, "width": w}, features=_clustered_events(300 + k, n, h, w),
                                                              timestamp_multiplier=1e6, convert_to_relative=True).features)).to(DEV)
          for k, n in enumerate((n1, n2))]
    for normalize in (True, False):
        singles = [voxelize_device(e, bins, h, w, normalize) for e in ev]
        pair = voxelize_pair_device(ev[0], ev[1], bins, h, w, normalize)
        for k in (0, 1):
            assert (pair[k] != 0).sum() == (singles[k] != 0).sum()
            assert maxerr(pair[k], singles[k]) < 2e-5 * max(1.0, float(singles[k].abs().max()))
    if n1 + n2 < 400_000:
        ref = O.voxelize(ev[1].cpu().numpy(), bins, h, w, normalize=True)
        np.testing.assert_allclose(voxelize_pair_device(ev[0], ev[1], bins, h, w, True)[1].cpu().numpy(), ref, atol=3e-5, rtol=1e-5)
    monkeypatch.setenv("EEM_VOX_DIRECT", "1")
    direct = voxelize_pair_device(ev[0], ev[1], bins, h, w, True)
    assert maxerr(direct[0], voxelize_device(ev[0], bins, h, w, True)) < 2e-4 and maxerr(direct[1], voxelize_device(ev[1], bins, h, w, True)) < 2e-4


def test_errors_are_loud():
    net, _ = make_net(1)
    with pytest.raises(AttributeError):
        net(torch.zeros(1, 5, 64, 64, device=DEV), torch.zeros(1, 5, 64, 64, device=DEV))
    net.change_imagesize((64, 64))
    with pytest.raises(_lib.EEMFlowHipError):
        net(torch.zeros(1, 5, 64, 64), torch.zeros(1, 5, 64, 64))
    with pytest.raises(_lib.EEMFlowHipError):      # padded 64x64 -> grid 1x1 ok; 16x16 raw pads to 64: ok; use bad ctx arg
        _lib.check(_lib.lib().eemflow_forward(net._context(torch.device(DEV)), None, None, 1, 64, 64, None, 64, 64, None))
    handle = ctypes.c_void_p()
    _lib.check(_lib.lib().eemflow_create(0, ctypes.byref(handle)))
    with pytest.raises(_lib.EEMFlowHipError, match="no weights"):
        x = torch.zeros(1, 5, 64, 64, device=DEV)
        o = torch.zeros(1, 2, 64, 64, device=DEV)
        _lib.check(_lib.lib().eemflow_forward(handle, x.data_ptr(), x.data_ptr(), 1, 64, 64, o.data_ptr(), 64, 64, None))
    _lib.lib().eemflow_destroy(handle)


@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (2, 260, 346), (1, 100, 150), (3, 192, 256)])
def test_ring_stride2_kernel_equals_the_chunked_one(monkeypatch, b, h, w):
    """conv_s2r.hip (8-wave blocks, input and weights through one LDS-DMA ring of k-step slices; pconv2_1 and pconv3_1; off by
    default - it measured slower - and switched on per launch by EEM_S2R=1) against the default kernels: the same sums in another
    order - stage tensors and flow to fp32 round-off, at the headline size, at ragged ones (border tiles, several tiles per block)
    and with a batch."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(56, b, h, w))
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("EEM_S2R", on)
        net, _ = make_net(45, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
        outs.append((flow, net.stage("f12").clone(), net.stage("f13").clone()))
    for a, c in zip(*outs):
        assert maxerr(a, c) < 2e-5 * max(1.0, float(c.abs().max()))
    assert float(outs[0][1].abs().max()) > 0.1


@pytest.mark.parametrize("form", ["f2", "f4", "default"])
@pytest.mark.parametrize("h,w", [(192, 320), (260, 346), (720, 1280)])
def test_winograd_encoder_equals_direct_convolution(monkeypatch, h, w, form):
    """The Winograd kernels - F(2x2,3x3) (EEM_WINO=2), F(4x4,3x3) on every stride-1 layer (EEM_WINO4_LAYERS=7), the default mix -
    and the enc1 kernel against the direct-convolution kernels of the same library (EEM_WINO=0 is read when weights are loaded,
    EEM_NO_ENC1 at launch): same stage tensors and flow to fp32 round-off."""
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(77, 1, h, w))

    def run():
        net, _ = make_net(78, graph=False)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(e1, e2)[1][0].clone()
        return flow, {k: net.stage(k).clone() for k in ("f11", "f12", "f13", "pool_1", "pool_2", "pool_3")}
    monkeypatch.delenv("EEM_WINO", raising=False)
    monkeypatch.delenv("EEM_WINO4_LAYERS", raising=False)
    if form == "f2":
        monkeypatch.setenv("EEM_WINO", "2")
    elif form == "f4":
        monkeypatch.setenv("EEM_WINO4_LAYERS", "7")
    fast_flow, fast = run()
    monkeypatch.delenv("EEM_WINO4_LAYERS", raising=False)
    monkeypatch.setenv("EEM_WINO", "0")
    monkeypatch.setenv("EEM_NO_ENC1", "1")
    ref_flow, ref = run()
    for k in fast:
        scale = max(1.0, float(ref[k].abs().max()))
        assert maxerr(fast[k], ref[k]) < 2e-5 * scale, k
    assert maxerr(fast_flow, ref_flow) < 2e-5


# ------------------------------------------------------------------------------------------------ coalesced frames (eemflow_forward_many)
@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("n,h,w", [(4, 720, 1280), (3, 260, 346), (5, 128, 192)])
def test_forward_many_equals_the_batched_forward(n, h, w, graph):
    """n independent samples of the evaluation loop (test_mvsec.py:580-597), each in its own tensors, through ONE batch-n chain:
    bitwise the batch-n forward of the stacked frames (same kernels, same launch configuration), every frame in its own flow tensor."""
    net, sd = make_net(41, graph=graph)
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(42, n, h, w))
    pad = [torch.empty(7 + 3 * i, device=DEV) for i in range(2 * n)]          # unrelated allocations between the frames' buffers
    frames = [(e1[i:i + 1].clone(), e2[i:i + 1].clone()) for i in range(n)]
    with torch.no_grad():
        ref = net(e1, e2)[1][0].clone()
        outs = net.forward_many(frames)
        f12_many = net.stage("f12").clone()
        _ = net(e1, e2)
        f12_ref = net.stage("f12")
    assert len(outs) == n and len(pad) == 2 * n
    for i, ((r1, r2), preds) in enumerate(outs):
        assert r1 is frames[i][0] and r2 is frames[i][1] and len(preds) == 1
        assert preds[0].shape == (1, 2, h, w)
        assert torch.equal(preds[0][0], ref[i]), f"frame {i}"
    assert torch.equal(f12_many, f12_ref)
    # a second call with OTHER buffers replays the same graph through a rewritten table
    frames2 = [(a.clone(), b.clone()) for a, b in reversed(frames)]
    with torch.no_grad():
        outs2 = net.forward_many(frames2)
    for i in range(n):
        assert torch.equal(outs2[i][1][0][0], ref[n - 1 - i])
    if graph:
        gs = (ctypes.c_longlong * 3)()
        _lib.check(_lib.lib().eemflow_graph_stats(net._ctx, ctypes.byref(gs)))
        assert gs[0] == 2, list(gs)          # one capture for the batch form, one for the per-frame form


def test_forward_many_against_the_oracle_per_frame():
    n, h, w = 4, 260, 346
    net, sd = make_net(43)
    net.change_imagesize((h, w))
    pairs = [tuple(torch.from_numpy(a) for a in synthetic_voxel_pair(50 + i, 1, h, w)) for i in range(n)]
    with torch.no_grad():
        outs = net.forward_many([(a.to(DEV), b.to(DEV)) for a, b in pairs])
        for i, (a, b) in enumerate(pairs):
            ref, _ = O.eemflow_forward(O.to_torch_sd(sd), a, b)
            assert maxerr(outs[i][1][0], ref) < FLOW_TOL, i


@pytest.mark.parametrize("h,w", [(720, 1280), (260, 346)])
def test_timed_configuration_against_the_oracle(h, w):
    """The launch configuration bench.py times: TEN distinct batch-1 samples per eemflow_forward_many call with
    the two-chain hint (frames_in_flight 2) - batch 10 => F(4x4,3x3) on every stride-1 layer, the column-order tile walk of the 64-channel
    layers, the multi-tile grouped decoder kernel - compared DIRECTLY with ten oracle forwards (EEMFlow.py:122-183; loop
    test_mvsec.py:580-597): every frame's flow and its f11 / f12 of both event volumes."""
    n = 10
    net, sd = make_net(46)
    net.frames_in_flight = 2
    net.change_imagesize((h, w))
    tsd = O.to_torch_sd(sd)
    pairs = [tuple(torch.from_numpy(a) for a in synthetic_voxel_pair(70 + i, 1, h, w)) for i in range(n)]
    with torch.no_grad():
        outs = net.forward_many([(a.to(DEV), b.to(DEV)) for a, b in pairs])
        f11, f12 = net.stage("f11").cpu(), net.stage("f12").cpu()
    assert f11.shape[0] == 2 * n and f12.shape[0] == 2 * n
    worst = 0.0
    for i, (a, b) in enumerate(pairs):
        with torch.no_grad():
            ref, st = O.eemflow_forward(tsd, a, b, keep=True)
        err = maxerr(outs[i][1][0], ref)
        worst = max(worst, err)
        assert err < FLOW_TOL, (i, err)
        # images 0..n-1 of a batched chain are the frames' first event volumes, n..2n-1 their second ones
        assert maxerr(f11[i:i + 1], st["f11"]) < FEAT_TOL and maxerr(f11[n + i:n + i + 1], st["f21"]) < FEAT_TOL, i
        assert maxerr(f12[i:i + 1], st["f12"]) < FEAT_TOL and maxerr(f12[n + i:n + i + 1], st["f22"]) < FEAT_TOL, i
    print(f"timed configuration {h}x{w}: worst flow error of {n} frames against the oracle {worst:.2e}")


def test_forward_many_rejects_what_it_cannot_run():
    net, _ = make_net(44)
    net.change_imagesize((128, 192))
    a = torch.zeros(1, 5, 128, 192, device=DEV)
    with pytest.raises(ValueError):
        net.forward_many([])
    with pytest.raises(ValueError):
        net.forward_many([(a, a)] * 17)
    with pytest.raises(ValueError):
        net.forward_many([(torch.zeros(2, 5, 128, 192, device=DEV),) * 2])
    with pytest.raises(_lib.EEMFlowHipError):
        net.forward_many([(a.cpu(), a.cpu())])


def test_stage_f13_follows_every_graph_replay():
    """inference leaves f13 unwritten (pconv3_3 pools it in its epilogue) and eemflow_get_stage("f13") re-runs the layer
    with stores.  The flag that says so was only set at capture time: a replay with NEW inputs after a stage("f13") call then returned
    the PREVIOUS frame's f13.  Two different frames through one cached graph, stage("f13") after each, against the oracle."""
    h, w = 128, 192
    net, sd = make_net(45)
    net.change_imagesize((h, w))
    tsd = O.to_torch_sd(sd)
    for seed in (60, 61, 62):
        e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed, 1, h, w))
        with torch.no_grad():
            _ = net(e1.to(DEV), e2.to(DEV))
            _, st = O.eemflow_forward(tsd, e1, e2, keep=True)
        assert maxerr(net.stage("f13")[:1], st["f13"]) < FEAT_TOL, seed
    gs = (ctypes.c_longlong * 3)()
    _lib.check(_lib.lib().eemflow_graph_stats(net._ctx, ctypes.byref(gs)))
    assert gs[0] == 1 and gs[1] == 3, list(gs)


# ------------------------------------------------------------------------------------------------ pconv1_1 inside pconv1_2's block
@pytest.mark.parametrize("b,h,w", [(1, 720, 1280), (2, 256, 384), (1, 192, 1000), (3, 100, 152), (1, 64, 64)])
def test_fused_first_two_layers_equal_the_two_launches(b, h, w, monkeypatch):
    """conv_enc12.hip computes pconv1_1 (EEMFlow.py:75) inside pconv1_2's block (EEMFlow.py:76) with the arithmetic of the two kernels it
    replaces (same MFMA operand order, same k order): f11, the pooled map, the flow and `a1` - re-run on demand, since it is never
    written - are BITWISE those of the two launches; and both agree with the oracle.  Sizes: the headline, tiles cut by the right and
    bottom edges, replicate-padded rows, an image smaller than one tile."""
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(72, b, h, w))
    d1, d2 = e1.to(DEV), e2.to(DEV)                              # alive while stage("a1") re-runs the first layer on them
    got = {}
    for off in ("1", "0"):
        monkeypatch.setenv("EEM_FUSE12", "0" if off == "1" else "1")
        net, sd = make_net(71)
        net.change_imagesize((h, w))
        with torch.no_grad():
            flow = net(d1, d2)[1][0]
        got[off] = {"flow": flow.clone(), "f11": net.stage("f11").clone(), "pool_1": net.stage("pool_1").clone(), "a1": net.stage("a1").clone()}
        with torch.no_grad():                                    # a second frame through the cached graph, then a1 again
            flow2 = net(d2, d1)[1][0]
        got[off]["a1_swapped"] = net.stage("a1").clone()
        got[off]["flow_swapped"] = flow2.clone()
    for k in got["1"]:
        assert torch.equal(got["0"][k], got["1"][k]), k
    with torch.no_grad():
        ref, st = O.eemflow_forward(O.to_torch_sd(sd), e1, e2, keep=True)
    assert maxerr(got["0"]["f11"][:b], st["f11"]) < FEAT_TOL
    assert maxerr(got["0"]["flow"], ref) < FLOW_TOL


def test_voxelize_many_equals_the_single_calls():
    """eemflow_voxelize_many: 2 x n event sets (ragged lengths, one of a single event) in one launch sequence - every grid bitwise the
    grid eemflow_voxelize gives for that set alone (same kernels, the job index only selects the buffers), and one of them against
    the oracle's voxel grid."""
    from eemflow_amd.hrem import synthetic_hrem_events
    from eemflow_amd.voxelizer import voxelize_device, voxelize_many_device
    h, w, bins = 260, 346, 5
    sets = []
    for k, nev in enumerate([20000, 1, 7777, 50000, 333, 20000, 12345]):
        seq = EventSequence(None, {"height": h, "width": w}, features=synthetic_hrem_events(80 + k, nev, h, w),
                            timestamp_multiplier=1e6, convert_to_relative=True)
        sets.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(DEV))
    for normalize in (True, False):
        many = voxelize_many_device(sets, bins, h, w, normalize=normalize)
        for k, ev in enumerate(sets):
            assert torch.equal(many[k], voxelize_device(ev, bins, h, w, normalize=normalize)), (k, normalize)
    many = voxelize_many_device(sets, bins, h, w)
    for k in (3, 4):
        assert maxerr(many[k], O.voxelize(sets[k].cpu().numpy(), bins, h, w, normalize=True)) < 1e-4, k
    with pytest.raises(ValueError):
        voxelize_many_device(sets * 5, bins, h, w)


# ------------------------------------------------------------------------------------------------ normalisation left to pconv1_1
@pytest.mark.parametrize("nev", [200000, 2000000])
def test_deferred_normalisation_route(nev):
    """loader_utils.py:527-535 folded into its consumer: the voxelizer leaves RAW grids with the record {mean, sd, scale, any} behind
    them (normalize='deferred'), pconv1_1 normalises as it reads (EEMFlow.py:135).  The record against the oracle's statistics, the raw
    grid bitwise the normalize=False grid, and the flow of n frames against oracle-voxelize -> oracle-forward (<= 1e-4) and against the
    normalised-grid route of this library (round-off of one multiply per voxel)."""
    from eemflow_amd.hrem import synthetic_hrem_events
    from eemflow_amd.voxelizer import norm_record, voxelize_device, voxelize_many_device
    h, w, bins, n = 720, 1280, 5, 2
    net, sd = make_net(91)
    net.change_imagesize((h, w))
    sets = []
    for k in range(2 * n):
        seq = EventSequence(None, {"height": h, "width": w}, features=synthetic_hrem_events(92 + k, nev, h, w),
                            timestamp_multiplier=1e6, convert_to_relative=True)
        sets.append(torch.from_numpy(np.ascontiguousarray(seq.features)).to(DEV))
    raw = voxelize_many_device(sets, bins, h, w, normalize="deferred")
    normed = voxelize_many_device(sets, bins, h, w, normalize=True)
    for k in range(2 * n):
        assert torch.equal(raw[k], voxelize_device(sets[k], bins, h, w, normalize=False)), k
        g = raw[k][raw[k] != 0].double()
        rec = norm_record(raw[k]).cpu()
        assert abs(float(rec[0]) - float(g.mean())) < 1e-6 and abs(float(rec[1]) - float(g.std())) < 1e-6 * float(g.std()) + 1e-7
        assert rec[2] == 1 and rec[3] == 1
    with torch.no_grad():
        got = net.forward_many([(raw[2 * i][None], raw[2 * i + 1][None]) for i in range(n)], deferred_norm=True)
        ref2 = net.forward_many([(normed[2 * i][None], normed[2 * i + 1][None]) for i in range(n)])
        for i in range(n):
            assert maxerr(got[i][1][0], ref2[i][1][0]) < 2e-5, i
        v1 = torch.from_numpy(O.voxelize(sets[0].cpu().numpy(), bins, h, w, normalize=True))[None]
        v2 = torch.from_numpy(O.voxelize(sets[1].cpu().numpy(), bins, h, w, normalize=True))[None]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd), v1, v2)
    assert maxerr(got[0][1][0], ref) < FLOW_TOL
    # the record is a caller's contract: volumes without one are refused
    with pytest.raises(ValueError):
        net.forward_many([(normed[0][None].clone(), normed[1][None].clone())], deferred_norm=True)


def test_deferred_normalisation_edge_records():
    """One non-zero voxel (the reference's std is NaN: mean subtracted only -> 0), and a shape the 16-byte first-layer kernel does not take
    (MVSEC 346x260: the library refuses the deferred form there instead of computing something else)."""
    from eemflow_amd.voxelizer import norm_record, voxelize_device
    h, w, bins = 128, 192, 5
    ev = torch.tensor([[0.0, 5.0, 7.0, 1.0]], dtype=torch.float64, device=DEV)
    g = voxelize_device(ev, bins, h, w, normalize="deferred")
    rec = norm_record(g).cpu()
    assert rec[3] == 1 and rec[2] == 0 and float(rec[0]) == 1.0          # any, no scaling (sd is NaN for one voxel), mean = the vote
    net, sd = make_net(93)
    net.change_imagesize((h, w))
    z = voxelize_device(ev, bins, h, w, normalize="deferred")
    with torch.no_grad():
        got = net.forward_many([(g[None], z[None])], deferred_norm=True)[0][1][0]
        v = torch.from_numpy(O.voxelize(ev.cpu().numpy(), bins, h, w, normalize=True))[None]
        ref, _ = O.eemflow_forward(O.to_torch_sd(sd), v, v)
    assert maxerr(got, ref) < FLOW_TOL
    net2, _ = make_net(93)
    net2.change_imagesize((260, 346))
    ev2 = torch.tensor([[0.0, 5.0, 7.0, 1.0], [1.0, 9.0, 3.0, 0.0]], dtype=torch.float64, device=DEV)
    a = voxelize_device(ev2, bins, 260, 346, normalize="deferred")
    with pytest.raises(_lib.EEMFlowHipError), torch.no_grad():
        net2.forward_many([(a[None], a[None])], deferred_norm=True)


def test_batched_decoder_kernel_equals_the_one_tile_kernel(monkeypatch):
    """tail_conv_multi_kernel (five pixel tiles per block: the grouped 20 -> 20 decoder convs of a batched chain, EEMFlow.py:37-69) against
    tail_conv_kernel (EEM_NO_TAIL_MULTI=1): the same fragments in the same order - bitwise the same flow; batch 10 at the headline size,
    and a batch whose tile count is not a multiple of five."""
    for b, h, w in ((10, 720, 1280), (7, 260, 346)):
        e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(95, b, h, w))
        got = {}
        for off in ("1", "0"):
            monkeypatch.setenv("EEM_NO_TAIL_MULTI", off)
            net, sd = make_net(94)
            net.change_imagesize((h, w))
            with torch.no_grad():
                got[off] = (net(e1, e2)[1][0].clone(), net.stage("flowcat").clone())
        assert torch.equal(got["0"][1], got["1"][1]) and torch.equal(got["0"][0], got["1"][0]), (b, h, w)
        with torch.no_grad():                                  # both sizes against the oracle (the first and the last frame of the batch)
            for i in (0, b - 1):
                ref, _ = O.eemflow_forward(O.to_torch_sd(sd), e1[i:i + 1].cpu(), e2[i:i + 1].cpu())
                assert maxerr(got["0"][0][i:i + 1], ref) < FLOW_TOL, (b, h, w, i)
