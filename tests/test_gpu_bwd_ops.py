"""Backward operators on the GPU against torch autograd through the oracle's restatement of the reference
ops (the restatements are pinned to the reference by tests/test_oracle_golden.py).  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from eemflow_amd import ops_bwd
from oracle import eemflow_plus_oracle as P
from oracle import eraft_oracle as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("form", ["rows", "scatter"])
@pytest.mark.parametrize("b,c,h,w", [(1, 16, 16, 24), (2, 32, 20, 28)])
def test_corr_lookup_and_pyramid_bwd(monkeypatch, b, c, h, w, form):
    """form: the lookup adjoint as a gather over the rows of every pixel's 10 x 10 window (default) or as the per-tap atomic
    scatter (EEM_LOOKUP_BWD_SCATTER=1, read per call)."""
    monkeypatch.setenv("EEM_LOOKUP_BWD_SCATTER", "1" if form == "scatter" else "0")
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    f2 = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    coords = E.coords_grid(b, h, w) + 3.0 * torch.randn(b, 2, h, w, generator=g)      # fractional, partly out of range
    pyr = E.corr_pyramid(f1, f2)
    for p in pyr:
        p.retain_grad()
    out = E.corr_lookup(pyr, coords)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    d = ops_bwd.corr_lookup_bwd(coords.to(DEV), dout.to(DEV))
    # level gradients BEFORE the pooling chain: autograd's .grad of level l includes the contributions folded in from
    # coarser levels, so compare the direct scatter of the coarsest level and the folded result of level 0
    assert rel(d[3], pyr[3].grad) < 1e-5
    d1, d2 = ops_bwd.corr_pyramid_bwd(f1.detach().to(DEV), f2.detach().to(DEV), d)
    assert rel(d[0], pyr[0].grad) < 1e-5                     # folded in place
    assert rel(d1, f1.grad) < 1e-4 and rel(d2, f2.grad) < 1e-4


def test_convex_upsample_bwd():
    g = torch.Generator().manual_seed(6)
    b, h, w = 2, 9, 13
    flow = torch.randn(b, 2, h, w, generator=g, requires_grad=True)
    mask = torch.randn(b, 576, h, w, generator=g, requires_grad=True)
    out = E.convex_upsample(flow, mask)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    dflow, dmask = ops_bwd.convex_upsample_bwd(flow.detach().to(DEV), mask.detach().to(DEV), dout.to(DEV))
    assert rel(dflow, flow.grad) < 1e-5 and rel(dmask, mask.grad) < 1e-5


@pytest.mark.parametrize("mode,fn", [(0, P.warp_align_true), (1, P.torch_warp), (2, P.warping_layer_no_div)])
def test_warp_bwd(mode, fn):
    g = torch.Generator().manual_seed(7 + mode)
    b, c, h, w = 2, 5, 17, 23
    x = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    flo = (4.0 * torch.randn(b, 2, h, w, generator=g)).requires_grad_(True)           # reaches out of bounds
    if mode == 2:
        # WarpingLayer_no_div masks with grid_sample(ones) >= 1.0, which sits on a rounding edge for every interior sample
        # (the four weights sum to 1 +- 1 ulp; the reference's own mask changes with its CPU code path, see
        # test_gpu_plus.py).  The mask carries no gradient, so the adjoint is checked with the mask the GPU forward
        # kernel produces (the backward kernel evaluates the identical expression).
        import ctypes
        from eemflow_amd import _lib
        ones = torch.ones(b, 1, h, w, device=DEV)
        mk = torch.empty_like(ones)
        fl = flo.detach().to(DEV).contiguous()
        _lib.check(_lib.lib().eemplus_warp(ones.data_ptr(), fl.data_ptr(), b, 1, h, w, 2, mk.data_ptr(),
                                           _lib.current_stream_ptr(torch.device(DEV))))
        out = P.torch_warp(x, flo) * (mk.cpu() != 0).float()
    else:
        out = fn(x, flo)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    dx, dflow = ops_bwd.warp_bwd(x.detach().to(DEV), flo.detach().to(DEV), dout.to(DEV), mode)
    assert rel(dx, x.grad) < 1e-5
    assert rel(dflow, flo.grad) < 1e-4


def test_bwd_ops_reject_cpu_tensors():
    with pytest.raises(Exception):
        ops_bwd.warp_bwd(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 4, 4), torch.zeros(1, 1, 4, 4), 0)


@pytest.mark.parametrize("cin,cout,h,w,b", [(176, 8, 72, 96, 1), (184, 3, 90, 160, 2), (32, 2, 64, 65, 1), (256, 2, 60, 80, 1), (67, 5, 65, 70, 1),
                                           (100, 2, 30, 41, 2), (200, 1, 33, 47, 1), (64, 2, 20, 20, 3)])   # (the last three: the 32-pixel form, ragged)
def test_few_output_conv_vs_torch_and_generic_kernel(monkeypatch, cin, cout, h, w, b):
    """Layers of <= 8 output channels at >= 4096 pixels (EEMFlow+'s mask estimator tail, model/cdc_model.py dense blocks) run as a
    direct convolution on the vector pipe (gconv.h: fewout_*); EEM_NO_FEWOUT=1 (read per call) keeps them on the matrix-core kernel."""
    from eemflow_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    x = torch.randn(b, cin, h, w, generator=g)
    ref = torch.nn.functional.leaky_relu(conv(x.double().float()), 0.1).detach()
    convd = conv.to(DEV)
    with torch.no_grad():
        few = ops.conv2d(convd, x.to(DEV), act=ops.ACT_LEAKY).cpu()
        monkeypatch.setenv("EEM_NO_FEWOUT", "1")
        gen = ops.conv2d(convd, x.to(DEV), act=ops.ACT_LEAKY).cpu()
    scale = float(ref.abs().max())
    assert float((few - ref).abs().max()) < 2e-5 * max(scale, 1.0) * (cin / 32) ** 0.5
    assert float((gen - ref).abs().max()) < 2e-5 * max(scale, 1.0) * (cin / 32) ** 0.5
    assert not torch.equal(few, gen) or cin < 8        # two kernels, two summation orders


def test_sixteen_output_conv_on_the_lds_tiled_kernel(monkeypatch):
    """160 -> 16 (EEMFlow+'s fourth estimator layer): one 16-cout tile, a tile row per wave (gconv16.hip, WM = 1) against torch and
    against the generic kernel (EEM_NO_GCONV16=1, read per call)."""
    from eemflow_amd import ops
    g = torch.Generator().manual_seed(16)
    conv = torch.nn.Conv2d(160, 16, 3, padding=1)
    x = torch.randn(2, 160, 96, 128, generator=g)
    ref = torch.nn.functional.leaky_relu(conv(x), 0.1).detach()
    convd = conv.to(DEV)
    with torch.no_grad():
        tiled = ops.conv2d(convd, x.to(DEV), act=ops.ACT_LEAKY).cpu()
        monkeypatch.setenv("EEM_NO_GCONV16", "1")
        gen = ops.conv2d(convd, x.to(DEV), act=ops.ACT_LEAKY).cpu()
    assert float((tiled - ref).abs().max()) < 1e-4 and float((gen - ref).abs().max()) < 1e-4
    assert not torch.equal(tiled, gen)


@pytest.mark.parametrize("cin,cout,k,h,w,b", [(2, 128, 7, 60, 80, 1), (1, 48, 7, 37, 53, 2), (2, 32, 3, 33, 40, 1)])
def test_one_pair_input_conv_with_taps_as_k_steps(monkeypatch, cin, cout, k, h, w, b):
    """Inputs of one channel pair (E-RAFT's flow through the 7x7 convf1, model/update.py:70) run with the taps as the k-steps of one
    batch (gconv.hip: gconv_taps_kernel); EEM_NO_TAPS_KERNEL=1 (read per call) keeps the generic kernel's one-k-step batches."""
    from eemflow_amd import ops
    g = torch.Generator().manual_seed(cin * 100 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
    x = torch.randn(b, cin, h, w, generator=g)
    ref = torch.relu(conv(x)).detach()
    convd = conv.to(DEV)
    with torch.no_grad():
        fast = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
        monkeypatch.setenv("EEM_NO_TAPS_KERNEL", "1")
        gen = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
    assert float((fast - ref).abs().max()) < 2e-5 and float((gen - ref).abs().max()) < 2e-5
    assert float((fast - gen).abs().max()) < 2e-5      # one pair: the same k order; several: taps within a pair instead of pairs within a tap


@pytest.mark.parametrize("cin,cout,k,h,w,b", [(336, 256, (1, 1), 60, 80, 1), (384, 128, (1, 5), 60, 80, 1), (384, 128, (5, 1), 60, 80, 1),
                                               (256, 192, (3, 3), 60, 80, 1), (128, 64, (3, 3), 60, 80, 1), (128, 256, (3, 3), 45, 64, 2)])
def test_lds_tiled_conv_with_two_k_groups(monkeypatch, cin, cout, k, h, w, b):
    """E-RAFT's update block at batch 1: launches of at most one block per CU run two groups of four waves per tile that split the
    channel chunks (gconv16.hip, KG = 2; 21 chunks: 11 + 10), with 3 / 5 / 6-row tiles chosen by rounds x rows.  Against torch and
    against the generic kernel (EEM_NO_GCONV16=1, read per launch).  (EEM_NO_GCONVB=1: the larger of these shapes would otherwise
    take the bf16-piece kernel, which has its own tests in test_gpu_eraft.py / test_gpu_plus.py.)"""
    from eemflow_amd import ops
    monkeypatch.setenv("EEM_NO_GCONVB", "1")
    g = torch.Generator().manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, padding=(k[0] // 2, k[1] // 2))
    x = torch.randn(b, cin, h, w, generator=g)
    ref = torch.relu(conv(x)).detach()
    convd = conv.to(DEV)
    with torch.no_grad():
        tiled = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
        monkeypatch.setenv("EEM_NO_GCONV16", "1")
        gen = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
    tol = 3e-5 * max(float(ref.abs().max()), 1.0) * (cin * k[0] * k[1] / 256) ** 0.5
    assert float((tiled - ref).abs().max()) < tol and float((gen - ref).abs().max()) < tol
    assert not torch.equal(tiled, gen)


def test_bf16_piece_wide_conv_and_an_infinity_in_the_input(monkeypatch):
    """gconvb.hip splits every operand into three bf16 pieces by a - (a & 0xffff0000): for a = inf that is inf - inf = NaN (ADVICE round 4
    item 5).  The documented contract (gconvb.hip header): the output is NON-FINITE at exactly the positions where the fp32 kernel's
    (gconv16, EEM_NO_GCONVB=1) is - NaN in place of its infinities - and every other value agrees; no non-finite input is ever turned
    into a number."""
    from eemflow_amd import ops
    g = torch.Generator().manual_seed(77)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1)
    x = torch.randn(2, 128, 48, 64, generator=g)
    x[0, 5, 10, 20] = float("inf")
    x[1, 100, 40, 3] = float("-inf")
    convd = conv.to(DEV)
    with torch.no_grad():
        monkeypatch.setenv("EEM_GCONVB_MINBLK", "1")
        fast = ops.conv2d(convd, x.to(DEV), act=ops.ACT_NONE).cpu()
        monkeypatch.setenv("EEM_NO_GCONVB", "1")
        plain = ops.conv2d(convd, x.to(DEV), act=ops.ACT_NONE).cpu()
    bad_f, bad_p = ~torch.isfinite(fast), ~torch.isfinite(plain)
    assert int(bad_p.sum()) == 2 * 9 * 128 and torch.equal(bad_f, bad_p)          # the 3 x 3 footprint of each infinity, every cout
    assert not torch.equal(fast[~bad_p], plain[~bad_p])                           # (the switch did switch)
    assert float((fast[~bad_p] - plain[~bad_p]).abs().max()) < 1e-4


@pytest.mark.parametrize("cin,cout,k,h,w,b", [(64, 96, 3, 48, 64, 1), (64, 96, 1, 48, 64, 1), (96, 128, 3, 120, 160, 2), (96, 128, 1, 120, 160, 2),
                                               (64, 96, 3, 46, 68, 1), (32, 64, 3, 90, 160, 2)])
def test_lds_tiled_conv_at_stride_two(monkeypatch, cin, cout, k, h, w, b):
    """The encoders' downsampling convs (E-RAFT model/extractor.py layer2 / layer3: 3x3 stride 2 and the 1x1 stride-2 shortcut) on
    gconv16's stride-2 form - a (TH - 1) * 2 + k row tile in LDS, B fragments from every second column - against torch and against
    the generic kernel (EEM_NO_G16_S2=1, read per call).  46 x 68: odd tile counts and a ragged right / bottom edge."""
    from eemflow_amd import ops
    g = torch.Generator().manual_seed(cin + cout + k)
    conv = torch.nn.Conv2d(cin, cout, k, stride=2, padding=k // 2)
    x = torch.randn(b, cin, h, w, generator=g)
    ref = torch.relu(conv(x)).detach()
    convd = conv.to(DEV)
    with torch.no_grad():
        tiled = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
        monkeypatch.setenv("EEM_NO_G16_S2", "1")
        gen = ops.conv2d(convd, x.to(DEV), act=ops.ACT_RELU).cpu()
    tol = 3e-5 * max(float(ref.abs().max()), 1.0) * (cin * k * k / 256) ** 0.5
    assert tiled.shape == ref.shape
    assert float((tiled - ref).abs().max()) < tol and float((gen - ref).abs().max()) < tol
    assert not torch.equal(tiled, gen) or k == 1       # (the switch did switch; a 1x1 over 64 channels can sum in the same order)


@pytest.mark.parametrize("form", ["gather", "scatter"])
@pytest.mark.parametrize("n,h,w,oh,ow", [(2, 6, 8, 512, 768), (1, 64, 96, 128, 192), (2, 5, 7, 33, 50), (1, 1, 9, 17, 40), (1, 12, 20, 12, 20)])
def test_bilinear_resize_adjoint(monkeypatch, form, n, h, w, oh, ow):
    """Adjoint of F.interpolate(bilinear, align_corners=True) (cdc_utils.py:83; the five full-resolution predictions of
    EEMFlow+.py:231-232): a wave per source pixel gathers its footprint of outputs (default), or four atomics per output pixel
    (EEM_RESIZE_BWD_SCATTER=1, read per call).  Against torch autograd."""
    import ctypes
    from eemflow_amd import _lib
    monkeypatch.setenv("EEM_RESIZE_BWD_SCATTER", "1" if form == "scatter" else "0")
    g = torch.Generator().manual_seed(h * 100 + ow)
    x = torch.randn(n, 2, h, w, generator=g, requires_grad=True)
    dout = torch.randn(n, 2, oh, ow, generator=g)
    torch.nn.functional.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True).backward(dout)
    dx = torch.empty(n, 2, h, w, device=DEV)
    d = dout.to(DEV)
    _lib.check(_lib.lib().eemop_resize_ac_bwd(d.data_ptr(), dx.data_ptr(), n * 2, h, w, oh, ow, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert rel(dx, x.grad) < 2e-5


def _conv_wgrad(x, dy, cin_total, ci0, cout, kh, kw, stride, ph, pw):
    """eemop_conv2d_bwd_weight on device tensors: (dw [cout][cin_total][kh][kw] with only the slice [ci0, ci0 + cic) written, db)."""
    from eemflow_amd import _lib
    n, cic, hin, win = x.shape
    dw = torch.zeros(cout, cin_total, kh, kw, device=DEV)
    db = torch.zeros(cout, device=DEV)
    _lib.check(_lib.lib().eemop_conv2d_bwd_weight(x.data_ptr(), dy.data_ptr(), n, hin, win, cin_total, ci0, cic, cout, kh, kw, stride, ph, pw,
                                                  dw.data_ptr(), db.data_ptr(), _lib.current_stream_ptr(torch.device(DEV))))
    torch.cuda.synchronize()
    return dw.cpu(), db.cpu()


@pytest.mark.parametrize("cin,cout,k,stride,n,h,w", [
    (16, 16, (3, 3), 1, 3, 160, 192),     # pconv1_2 at C3's map (Cfg16x16: whole rows of 192, eight K parts)
    (16, 16, (3, 3), 1, 2, 50, 640),      # ... C4's width: strips
    (16, 32, (3, 3), 2, 3, 160, 192),     # pconv2_1 (stride 2: two new input rows per slice)
    (32, 32, (3, 3), 1, 3, 80, 96),       # pconv2_2
    (32, 32, (3, 3), 1, 1, 37, 320),      # ... strips of 80, a segment count that does not divide the rows
    (32, 64, (3, 3), 2, 2, 80, 96),       # pconv3_1
    (64, 64, (3, 3), 1, 3, 40, 48),       # pconv3_2
    (64, 64, (3, 3), 1, 1, 96, 160),      # ... C4's map: two strips of 80
    (384, 128, (1, 5), 1, 2, 60, 80),     # SepConvGRU convz1 (model/update.py:36): six input chunks, two cout chunks
    (384, 128, (5, 1), 1, 2, 60, 80),     # convz2 (:40): strips of 40
    (256, 192, (3, 3), 1, 1, 60, 80),     # convc2 (:67)
    (128, 126, (3, 3), 1, 2, 30, 44),     # conv (:71): 126 couts = 64 + 62; a width that is 4 mod 8
    (96, 80, (3, 3), 1, 1, 20, 36),       # channel remainders on both sides (96 = 64 + 32, 80 = 64 + 16)
    (64, 96, (3, 3), 2, 2, 120, 160),     # the encoders' downsampling conv (model/extractor.py:13, layer2)
    (96, 128, (3, 3), 2, 1, 62, 84),      # ... layer3, odd output extents
])
def test_weight_gradient_ring_kernel(monkeypatch, cin, cout, k, stride, n, h, w):
    """wgrad_ring.hip (round 6: LDS rings of G slices and X rows that run ahead of the MFMAs, one 8-wave block per run of output rows)
    against torch autograd in float64 (conv2d's weight / bias gradient, model/update.py:33-60, EEMFlow.py:75-82 under
    train_mvsec.py:253-258) and against the kernel it replaces (EEM_NO_WGRAD_RING=1, read per call): <= 2e-5 of the largest gradient.
    Every block configuration, strips, ragged segments, channel remainders, stride 2, an input-channel slice of a wider weight."""
    g = torch.Generator().manual_seed(cin * 7 + cout + k[0])
    kh, kw = k
    ph, pw = kh // 2, kw // 2
    x = torch.randn(n, cin, h, w, generator=g)
    hout, wout = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    dy = torch.randn(n, cout, hout, wout, generator=g)
    wt = torch.zeros(cout, cin, kh, kw, dtype=torch.float64, requires_grad=True)
    bs = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    out = torch.nn.functional.conv2d(x.double(), wt, bs, stride=stride, padding=(ph, pw))
    out.backward(dy.double())
    xd, dyd = x.to(DEV), dy.to(DEV)
    monkeypatch.delenv("EEM_NO_WGRAD_RING", raising=False)
    monkeypatch.setenv("EEM_WGRAD_RING", "all")              # (by default only the shapes where it is the faster kernel take it)
    dw_r, db_r = _conv_wgrad(xd, dyd, cin, 0, cout, kh, kw, stride, ph, pw)
    monkeypatch.setenv("EEM_NO_WGRAD_RING", "1")
    dw_o, db_o = _conv_wgrad(xd, dyd, cin, 0, cout, kh, kw, stride, ph, pw)
    scale = float(wt.grad.abs().max())
    assert rel(dw_r, wt.grad) < 2e-5 and rel(db_r, bs.grad) < 2e-5, (rel(dw_r, wt.grad), rel(db_r, bs.grad))
    assert float((dw_r - dw_o).abs().max()) < 2e-5 * scale and not torch.equal(dw_r, dw_o)      # (the switch did switch)
    assert float((db_r - db_o).abs().max()) < 2e-5 * float(bs.grad.abs().max())
    # ... and the tile kernel's two arithmetic forms: products as six bf16-piece MFMAs (default from 32 couts up) against fp32 MFMAs
    # (EEM_NO_WGRAD_BX3=1, read per call) - the same sums to a few ulp of the largest one
    monkeypatch.setenv("EEM_NO_WGRAD_BX3", "1")
    dw_f, db_f = _conv_wgrad(xd, dyd, cin, 0, cout, kh, kw, stride, ph, pw)
    monkeypatch.delenv("EEM_NO_WGRAD_BX3", raising=False)
    assert rel(dw_o, wt.grad) < 2e-5 and rel(dw_f, wt.grad) < 2e-5 and rel(db_o, bs.grad) < 2e-5
    assert float((dw_f - dw_o).abs().max()) < 2e-5 * scale
    # an input-channel slice of a wider weight tensor (the GRU's [h | x] inputs, ops.Conv2d.backward: one call per input segment)
    monkeypatch.delenv("EEM_NO_WGRAD_RING", raising=False)
    lo = 16 if cin >= 48 else 0
    cic = cin - lo - (16 if cin >= 64 else 0)
    dw_s, _ = _conv_wgrad(xd[:, lo:lo + cic].contiguous(), dyd, cin, lo, cout, kh, kw, stride, ph, pw)
    assert float((dw_s[:, lo:lo + cic] - wt.grad[:, lo:lo + cic].float()).abs().max()) < 2e-5 * scale
    rest = torch.ones(cin, dtype=torch.bool)
    rest[lo:lo + cic] = False
    assert float(dw_s[:, rest].abs().max() if rest.any() else 0.0) == 0.0                         # nothing outside the slice is touched


@pytest.mark.parametrize("cs,cout,k,n,h,w", [((128, 128, 128), 128, (1, 5), 2, 60, 80), ((128, 128, 128), 128, (5, 1), 2, 60, 80),
                                              ((192, 64), 126, (3, 3), 1, 60, 80), ((128, 256), 128, (5, 1), 1, 36, 44),
                                              ((96, 48, 16), 96, (3, 3), 1, 24, 32)])
@pytest.mark.parametrize("ring", ["all", "none", "default"])
def test_weight_gradient_of_concatenated_inputs(monkeypatch, cs, cout, k, n, h, w, ring):
    """eemop_conv2d_bwd_weight_cat: the weight / bias gradient of a conv whose input is a torch.cat of up to three tensors
    (model/update.py:44,51 hx = cat([h, x]); :79 cat([out, flow])) in ONE call - on the ring kernel all segments ride one launch (a block's
    64-channel input chunk is looked up in the segment that holds it; remainders at segment ends) - against torch autograd in float64,
    with the ring kernel forced (EEM_WGRAD_RING=all), refused (none: one launch per segment on the old kernels) and by the default policy."""
    from eemflow_amd import _lib
    if ring == "default":
        monkeypatch.delenv("EEM_WGRAD_RING", raising=False)
    else:
        monkeypatch.setenv("EEM_WGRAD_RING", ring)
    g = torch.Generator().manual_seed(sum(cs) + cout)
    kh, kw = k
    ph, pw = kh // 2, kw // 2
    xs = [torch.randn(n, c, h, w, generator=g) for c in cs]
    dy = torch.randn(n, cout, h, w, generator=g)
    cin = sum(cs)
    wt = torch.zeros(cout, cin, kh, kw, dtype=torch.float64, requires_grad=True)
    bs = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(torch.cat(xs, 1).double(), wt, bs, padding=(ph, pw)).backward(dy.double())
    xd = [x.to(DEV) for x in xs]
    dyd = dy.to(DEV)
    dw = torch.zeros(cout, cin, kh, kw, device=DEV)
    db = torch.zeros(cout, device=DEV)
    px = [x.data_ptr() for x in xd] + [None] * (3 - len(xd))
    pc = list(cs) + [0] * (3 - len(cs))
    _lib.check(_lib.lib().eemop_conv2d_bwd_weight_cat(px[0], pc[0], px[1], pc[1], px[2], pc[2], dyd.data_ptr(), n, h, w, cout, kh, kw, 1, ph, pw,
                                                      dw.data_ptr(), db.data_ptr(), _lib.current_stream_ptr(torch.device(DEV))))
    torch.cuda.synchronize()
    assert rel(dw, wt.grad) < 2e-5 and rel(db, bs.grad) < 2e-5, (rel(dw, wt.grad), rel(db, bs.grad))


@pytest.mark.parametrize("cin,cout,k,n,h,w", [(64, 96, 3, 2, 240, 320), (96, 128, 3, 3, 120, 160), (64, 96, 1, 2, 240, 320), (96, 128, 1, 1, 120, 160),
                                               (64, 96, 3, 1, 46, 72), (80, 128, 3, 1, 34, 40), (96, 128, 1, 2, 30, 24)])
def test_stride_two_data_gradient_of_the_wide_layers(monkeypatch, cin, cout, k, n, h, w):
    """dgrad_s2w_kernel (round 6): the data gradient of the encoders' downsampling convs (model/extractor.py:13 3x3 stride 2; :33-36 the 1x1
    stride-2 shortcut) as four parity classes of dX computed from dY - against torch autograd in float64 and against the generic kernel's
    per-tap parity test (EEM_NO_DGRAD_S2W=1, read per call).  Ragged tiles, a channel count that is not a multiple of 16, odd output extents
    are refused (the generic kernel takes them)."""
    from eemflow_amd import _lib
    g = torch.Generator().manual_seed(cin + cout + k)
    pad = k // 2
    hout, wout = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
    dy = torch.randn(n, cout, hout, wout, generator=g)
    x = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt.double(), None, stride=2, padding=pad).backward(dy.double())
    wd, dyd = wt.to(DEV), dy.to(DEV)

    def run():
        dx = torch.full((n, cin, h, w), float("nan"), device=DEV)
        _lib.check(_lib.lib().eemop_conv2d_bwd_data(dyd.data_ptr(), wd.data_ptr(), n, h, w, cin, 0, cin, cout, k, k, 2, pad, pad, dx.data_ptr(),
                                                    _lib.current_stream_ptr(torch.device(DEV))))
        torch.cuda.synchronize()
        return dx.cpu()
    monkeypatch.delenv("EEM_NO_DGRAD_S2W", raising=False)
    fast = run()
    monkeypatch.setenv("EEM_NO_DGRAD_S2W", "1")
    gen = run()
    assert rel(fast, x.grad) < 2e-5 and rel(gen, x.grad) < 2e-5, (rel(fast, x.grad), rel(gen, x.grad))
    assert not torch.equal(fast, gen) or k == 1


@pytest.mark.parametrize("cin,cout,n,h,w", [(256, 2, 4, 60, 80), (32, 2, 2, 45, 64), (176, 8, 1, 24, 40), (40, 3, 2, 17, 36)])
def test_weight_gradient_of_the_few_output_layers(monkeypatch, cin, cout, n, h, w):
    """wgrad_few_kernel (round 6): weight + bias gradient of the 3x3 convs with <= 8 couts (E-RAFT's flow head 256 -> 2, model/update.py:10;
    EEMFlow+'s 32 -> 2 flow convs and 176 -> 8 mask estimator) on the vector pipe, against torch autograd in float64 and against the
    matrix-core kernel it replaces there (EEM_NO_WGRAD_FEW=1, read per call); an input-channel slice of a wider weight."""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bs = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wt, bs, padding=1).backward(dy.double())
    xd, dyd = x.to(DEV), dy.to(DEV)
    monkeypatch.delenv("EEM_NO_WGRAD_FEW", raising=False)
    dw_f, db_f = _conv_wgrad(xd, dyd, cin, 0, cout, 3, 3, 1, 1, 1)
    monkeypatch.setenv("EEM_NO_WGRAD_FEW", "1")
    dw_o, db_o = _conv_wgrad(xd, dyd, cin, 0, cout, 3, 3, 1, 1, 1)
    assert rel(dw_f, wt.grad) < 2e-5 and rel(db_f, bs.grad) < 2e-5 and rel(dw_o, wt.grad) < 2e-5
    assert not torch.equal(dw_f, dw_o)
    monkeypatch.delenv("EEM_NO_WGRAD_FEW", raising=False)
    lo, cic = 8, cin - 16
    dw_s, _ = _conv_wgrad(xd[:, lo:lo + cic].contiguous(), dyd, cin, lo, cout, 3, 3, 1, 1, 1)
    assert rel(dw_s[:, lo:lo + cic], wt.grad[:, lo:lo + cic]) < 2e-5
    assert float(dw_s[:, :lo].abs().max()) == 0.0 and float(dw_s[:, lo + cic:].abs().max()) == 0.0
