"""EEMFlow+ (EEMFlow_cdc) on MI355X behind the reference's nn.Module interface.

Mirrors model/EEMFlow/EEMFlow+.py:74-234 and model/EEMFlow/cdc_utils.py (cdc_model, conv): same constructor,
`change_imagesize`, `forward(events1, events2) -> ((events1, events2), [5 flows coarse -> fine])` and the same
136-tensor state_dict (including the parameters the reference registers but never uses: up3..up6,
cdc_model.upsample_output_conv, conv_1x1.0/.1).  Modules only hold parameters; the arithmetic runs in
libeemflow_hip.so.  CUDA tensors only.  Without gradients: `eemplus_forward`, the fused inference schedule.  Under autograd
(train_mvsec.py:245-258 on this model): the forward is assembled from the operator-level Functions of eemflow_amd/ops.py
(`_forward_ops`), including the in-place doubling upsample2d_flow_as applies to the coarser flow (as a functional rewrite).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .eemflow import convrelu
from .padder import InputPadder
from .weights import CORR_TAPS_53


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding, bias=True)


def conv(in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True):
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, dilation=dilation,
                        padding=((kernel_size - 1) * dilation) // 2, bias=True)]
    if isReLU:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
    return nn.Sequential(*layers)


class Decoder(nn.Module):
    """Parameter container mirroring EEMFlow+.py:38-49 (width 96)."""

    def __init__(self, in_channels, groups):
        super().__init__()
        self.in_channels = in_channels
        self.groups = groups
        self.conv1 = convrelu(in_channels, 96, 3, 1)
        self.conv2 = convrelu(96, 96, 3, 1, groups=groups)
        self.conv3 = convrelu(96, 96, 3, 1, groups=groups)
        self.conv4 = convrelu(96, 96, 3, 1, groups=groups)
        self.conv5 = convrelu(96, 64, 3, 1)
        self.conv6 = convrelu(64, 32, 3, 1)
        self.conv7 = nn.Conv2d(32, 2, 3, 1, 1)


class _DenseEstimator(nn.Module):
    def __init__(self, ch_in, f_channels=(128, 128, 96, 64, 32), ch_out=2):
        super().__init__()
        n = ch_in
        for i, f in enumerate(f_channels, start=1):
            setattr(self, f"conv{i}", conv(n, f))
            n += f
        self.num_feature_channel = n
        self.conv_last = conv(n, ch_out, isReLU=False)


class cdc_model(nn.Module):  # noqa: N801  (reference class name)
    def __init__(self):
        super().__init__()
        self.dense_estimator_mask = _DenseEstimator(64, f_channels=(32, 32, 32, 16, 8), ch_out=3)
        self.upsample_output_conv = nn.Sequential(conv(3, 16, kernel_size=3, stride=1, dilation=1), conv(16, 16, stride=2),
                                                  conv(16, 32, kernel_size=3, stride=1, dilation=1), conv(32, 32, stride=2))


class EEMFlow_cdc(nn.Module):  # noqa: N801
    def __init__(self, config, groups=3, n_first_channels=15, args=None):
        super().__init__()
        self.args = args
        self.groups = groups
        self.n_first_channels = n_first_channels
        self.pconv1_1 = convrelu(n_first_channels, 16, 3, 2)
        self.pconv1_2 = convrelu(16, 16, 3, 1)
        self.pconv2_1 = convrelu(16, 32, 3, 2)
        self.pconv2_2 = convrelu(32, 32, 3, 1)
        self.pconv2_3 = convrelu(32, 32, 3, 1)
        self.pconv3_1 = convrelu(32, 64, 3, 2)
        self.pconv3_2 = convrelu(64, 64, 3, 1)
        self.pconv3_3 = convrelu(64, 64, 3, 1)
        self.index = torch.tensor(CORR_TAPS_53)
        self.rconv2 = convrelu(32, 32, 3, 1)
        self.rconv3 = convrelu(64, 32, 3, 1)
        self.rconv4 = convrelu(64, 32, 3, 1)
        self.rconv5 = convrelu(64, 32, 3, 1)
        self.rconv6 = convrelu(64, 32, 3, 1)
        self.up3 = deconv(2, 2)
        self.up4 = deconv(2, 2)
        self.up5 = deconv(2, 2)
        self.up6 = deconv(2, 2)
        self.decoder2 = Decoder(87, groups)
        self.decoder3 = Decoder(87, groups)
        self.decoder4 = Decoder(87, groups)
        self.decoder5 = Decoder(87, groups)
        self.decoder6 = Decoder(87, groups)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self.cdc_model = cdc_model()
        self.conv_1x1 = nn.ModuleList([conv(15, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(16, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(32, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(64, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(64, 32, kernel_size=1, stride=1, dilation=1),
                                       conv(64, 32, kernel_size=1, stride=1, dilation=1)])
        self._ctx = None
        self._ctx_device = None
        self._weights_version = None

    def change_imagesize(self, img_size):
        self.image_size = img_size
        self.image_padder = InputPadder(img_size, mode='chairs', eval_pad_rate=64)

    def replicate(self, frames_in_flight=None):
        """A second module with the same weights, device, image size and mode and a context of its own: what keeps one more frame
        in flight on another HIP stream (harness.TestRaftEvents(frames_in_flight=...))."""
        twin = EEMFlow_cdc("", groups=self.groups, n_first_channels=self.n_first_channels, args=self.args)
        twin.load_state_dict(self.state_dict())
        twin = twin.to(next(self.parameters()).device)
        if hasattr(self, "image_size"):
            twin.change_imagesize(self.image_size)
        twin.train(self.training)
        twin.frames_in_flight = getattr(self, "frames_in_flight", 1) if frames_in_flight is None else frames_in_flight
        return twin

    def _flat_weights(self):
        return torch.cat([v.detach().reshape(-1).to(torch.float32).cpu() for v in self.state_dict().values()])

    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _context(self, device):
        L = _lib.lib()
        if self._ctx is None or self._ctx_device != device:
            self._release()
            handle = ctypes.c_void_p()
            _lib.check(L.eemplus_create(device.index if device.index is not None else torch.cuda.current_device(),
                                        ctypes.byref(handle)))
            self._ctx, self._ctx_device, self._weights_version = handle, device, None
        fp = self._fingerprint()
        if fp != self._weights_version:
            flat = self._flat_weights().contiguous()
            _lib.check(L.eemplus_load_weights(self._ctx, flat.data_ptr(), flat.numel(), self.n_first_channels, self.groups))
            self._weights_version = fp
        _lib.check(L.eemplus_set_frames_in_flight(self._ctx, max(1, int(getattr(self, "frames_in_flight", 1)))))
        return self._ctx

    def forward(self, events1, events2):
        if not (events1.is_cuda and events2.is_cuda):
            raise _lib.EEMFlowHipError("EEMFlow_cdc.forward: inputs must be CUDA (ROCm) tensors - there is no CPU path")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        e1, e2 = events1.contiguous().float(), events2.contiguous().float()
        if e1.shape != e2.shape or e1.dim() != 4 or e1.shape[1] != self.n_first_channels:
            raise ValueError(f"expected two (B,{self.n_first_channels},H,W) tensors")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return (events1, events2), self._forward_ops(e1, e2)
        b, _, h, w = e1.shape
        ctx = self._context(e1.device)
        out = torch.empty(5, b, 2, h, w, device=e1.device, dtype=torch.float32)
        padc = (ctypes.c_int * 4)(*self.image_padder._pad)
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().eemplus_forward(ctx, e1.data_ptr(), e2.data_ptr(), b, h, w, ctypes.byref(padc), out.data_ptr(),
                                                  _lib.current_stream_ptr(e1.device)))
        return (events1, events2), [out[i] for i in range(5)]

    MAX_COALESCE = 16

    def forward_many(self, frames):
        """Several INDEPENDENT samples of the evaluation loop (test_mvsec.py:580-597: one `model(events1, events2)` per sample at batch 1)
        as one batch-n chain of launches, each sample staying in its own tensors: `frames` is a sequence of (events1, events2) pairs of
        [1, C, H, W] tensors; returns one `((events1, events2), [flow6 .. flow2 at full resolution])` per sample, bitwise what `forward`
        gives for the samples stacked into one batch.  Inference only."""
        frames = list(frames)
        if not 1 <= len(frames) <= self.MAX_COALESCE:
            raise ValueError(f"forward_many: 1..{self.MAX_COALESCE} frames per call, got {len(frames)}")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        keep, shape = [], None
        for a, b in frames:
            if not (a.is_cuda and b.is_cuda):
                raise _lib.EEMFlowHipError("EEMFlow_cdc.forward_many: inputs must be CUDA (ROCm) tensors - there is no CPU path")
            a, b = a.contiguous().float(), b.contiguous().float()
            if a.shape != b.shape or a.dim() != 4 or a.shape[0] != 1 or a.shape[1] != self.n_first_channels:
                raise ValueError(f"forward_many: every frame is two (1,{self.n_first_channels},H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
            if shape is not None and a.shape != shape:
                raise ValueError("forward_many: all frames of a call share one shape")
            shape = a.shape
            keep.append((a, b))
        dev = keep[0][0].device
        h, w = int(shape[2]), int(shape[3])
        ctx = self._context(dev)
        n = len(keep)
        outs = [torch.empty(5, 1, 2, h, w, device=dev, dtype=torch.float32) for _ in range(n)]
        arr = ctypes.c_void_p * n
        p1, p2, po = arr(*[a.data_ptr() for a, _ in keep]), arr(*[b.data_ptr() for _, b in keep]), arr(*[o.data_ptr() for o in outs])
        padc = (ctypes.c_int * 4)(*self.image_padder._pad)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().eemplus_forward_many(ctx, n, p1, p2, h, w, padc, po, _lib.current_stream_ptr(dev)))
        return [((frames[i][0], frames[i][1]), [outs[i][k] for k in range(5)]) for i in range(n)]

    # ------------------------------------------------------------------ differentiable route (eemflow_amd/ops.py)
    def _lrelu_conv(self, seq, *xs):
        from . import ops
        return ops.conv2d(seq[0], *xs, act=ops.ACT_LEAKY)

    def _decoder_ops(self, dec, x):
        """Decoder.forward (EEMFlow+.py:60-71): grouped convs as one conv per group, then channel_shuffle."""
        from . import ops
        out = self._lrelu_conv(dec.conv1, x)
        G = self.groups
        for seq in (dec.conv2, dec.conv3, dec.conv4):
            conv = seq[0]
            if G == 1:
                out = self._lrelu_conv(seq, out)
                continue
            per_in, per_out = conv.weight.shape[1], conv.weight.shape[0] // G
            parts = [ops.conv2d(conv, ops.ChannelSlice.apply(out, g * per_in, per_in), act=ops.ACT_LEAKY,
                                weight=conv.weight[g * per_out:(g + 1) * per_out], bias=conv.bias[g * per_out:(g + 1) * per_out])
                     for g in range(G)]
            out = ops.ChannelShuffle.apply(ops.CatN.apply(*parts), G)
        out = self._lrelu_conv(dec.conv6, self._lrelu_conv(dec.conv5, out))
        return ops.conv2d(dec.conv7, out)

    def _cdc_from_init(self, flow_init, a, b):
        """cdc_model.forward after its upsampling (cdc_utils.py:160-173): flow_init (B,2,h,w) -> flow_up."""
        from . import ops
        est = self.cdc_model.dense_estimator_mask
        # (tensors with several consumers go through ops.fan_out: their gradients meet in one launch instead of autograd's chain of adds)
        fi_w, fi_t, fi_b = ops.fan_out(flow_init, 3)
        x = ops.CatN.apply(a, ops.Warp.apply(b, fi_w, 2))
        for i in range(1, 6):
            xa, xb = ops.fan_out(x, 2)
            x = ops.CatN.apply(self._lrelu_conv(getattr(est, f"conv{i}"), xa), xb)
        xo_f, xo_m = ops.fan_out(ops.conv2d(est.conv_last[0], x), 2)
        inter_flow = ops.ChannelSlice.apply(xo_f, 0, 2)
        ma, mb = ops.fan_out(ops.Sigmoid.apply(ops.ChannelSlice.apply(xo_m, 2, 1)), 2)
        m2 = ops.CatN.apply(ma, mb)                                        # the (B,1,h,w) mask broadcast over the two flow channels
        return ops.GRUBlend.apply(m2, ops.Warp.apply(fi_t, inter_flow, 1), fi_b)   # warp * (1 - m) + flow_init * m

    def _level_ops(self, l, f1l, f2l, flow_init):
        """One l-block of the forward (EEMFlow+.py:183-229) from cdc_model's upsampled flow_init -> (flow_up_l, flow_l)."""
        from . import ops
        f1a, f1c, f1r = ops.fan_out(f1l, 3)
        f2a, f2w = ops.fan_out(f2l, 2)
        a = self._lrelu_conv(self.conv_1x1[l], f1a)
        b = self._lrelu_conv(self.conv_1x1[l], f2a)
        fu_w, fu_c, fu_a, flow_up = ops.fan_out(self._cdc_from_init(flow_init, a, b), 4)
        cat = ops.CatN.apply(ops.LocalCorr53.apply(f1c, ops.Warp.apply(f2w, fu_w, 0)), self._lrelu_conv(getattr(self, f"rconv{l}"), f1r),
                             fu_c)
        return flow_up, ops.Add.apply(self._decoder_ops(getattr(self, f"decoder{l}"), cat), fu_a, 1)

    def _pyramid_ops(self, e1, e2):
        """Replicate pad, the shared encoder on [image1; image2] and three 2x2 poolings (EEMFlow+.py:162-175) -> f1[l], f2[l], l = 1..6."""
        from . import ops
        b, c, h, w = e1.shape
        pad = self.image_padder._pad
        x = torch.empty(2 * b, c, h + pad[2] + pad[3], w + pad[0] + pad[1], device=e1.device, dtype=torch.float32)
        ops.replicate_pad_into(e1, pad, x[:b])
        ops.replicate_pad_into(e2, pad, x[b:])
        feats = []
        y = x
        for name in ("pconv1_1", "pconv1_2", "pconv2_1", "pconv2_2", "pconv2_3", "pconv3_1", "pconv3_2", "pconv3_3"):
            y = self._lrelu_conv(getattr(self, name), y)
            if name in ("pconv1_2", "pconv2_3", "pconv3_3"):
                feats.append(y)
        for _ in range(3):
            feats.append(ops.AvgPool2.apply(feats[-1]))
        return {l: feats[l - 1][:b] for l in range(1, 7)}, {l: feats[l - 1][b:] for l in range(1, 7)}

    def _forward_ops(self, e1, e2):
        """EEMFlow_cdc.forward (EEMFlow+.py:158-234) as an autograd graph of HIP operators."""
        from . import ops
        if not (e1.is_cuda and all(p.is_cuda for p in self.parameters())):
            raise _lib.EEMFlowHipError("EEMFlow_cdc.forward: CUDA (ROCm) tensors required - there is no CPU path")
        b, _, h, w = e1.shape
        f1, f2 = self._pyramid_ops(e1, e2)
        zeros = torch.zeros(b, 2, f1[6].shape[2], f1[6].shape[3], device=e1.device, dtype=torch.float32)
        cat6 = ops.CatN.apply(ops.LocalCorr53.apply(f1[6], f2[6]), self._lrelu_conv(self.rconv6, f1[6]), zeros)
        flows = {6: self._decoder_ops(self.decoder6, cat6)}
        for l in (5, 4, 3, 2):
            hl, wl = f1[l].shape[-2:]
            if flows[l + 1].shape[-2:] != (hl, wl):
                # upsample2d_flow_as(if_rate=True) also scales its input in place (cdc_utils.py:85-86): continue with the scaled flow
                flow_init, flows[l + 1] = ops.UpsampleFlowAs.apply(flows[l + 1], hl, wl)
            else:
                flow_init = flows[l + 1]
            _, flows[l] = self._level_ops(l, f1[l], f2[l], flow_init)
        return [ops.UpsampleFlowAs.apply(flows[l], h, w)[0] for l in (6, 5, 4, 3, 2)]

    def level(self, l, flow_init):
        """Teacher-forced level l (5..2) on the feature pyramid of the last forward: cdc_model + warp + correlation + decoder from a
        supplied upsampled flow_init (B,2,h_l,w_l) -> (flow_up_l, flow_l).  EEMFlow+.py:184-193; for parity tests."""
        fi = flow_init.contiguous().float()
        up, out = torch.empty_like(fi), torch.empty_like(fi)
        with torch.cuda.device(fi.device):
            _lib.check(_lib.lib().eemplus_level(self._ctx, int(l), fi.data_ptr(), up.data_ptr(), out.data_ptr(),
                                                _lib.current_stream_ptr(fi.device)))
        return up, out

    def stage(self, name):
        L = _lib.lib()
        dims = (ctypes.c_int * 4)()
        _lib.check(L.eemplus_get_stage(self._ctx, name.encode(), None, 0, ctypes.byref(dims), None))
        out = torch.empty(*list(dims), device=self._ctx_device, dtype=torch.float32)
        with torch.cuda.device(self._ctx_device):
            _lib.check(L.eemplus_get_stage(self._ctx, name.encode(), out.data_ptr(), out.numel(), ctypes.byref(dims),
                                           _lib.current_stream_ptr(self._ctx_device)))
        return out

    def _release(self):
        if self._ctx is not None:
            _lib.lib().eemplus_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
