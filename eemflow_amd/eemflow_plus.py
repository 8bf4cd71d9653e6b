"""EEMFlow+ (EEMFlow_cdc) on MI355X behind the reference's nn.Module interface.

Mirrors model/EEMFlow/EEMFlow+.py:74-234 and model/EEMFlow/cdc_utils.py (cdc_model, conv): same constructor,
`change_imagesize`, `forward(events1, events2) -> ((events1, events2), [5 flows coarse -> fine])` and the same
136-tensor state_dict (including the parameters the reference registers but never uses: up3..up6,
cdc_model.upsample_output_conv, conv_1x1.0/.1).  Modules only hold parameters; the arithmetic runs in
libeemflow_hip.so (eemplus_* entry points).  Inference only; CUDA tensors only.
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
        return self._ctx

    def forward(self, events1, events2):
        if not (events1.is_cuda and events2.is_cuda):
            raise _lib.EEMFlowHipError("EEMFlow_cdc.forward: inputs must be CUDA (ROCm) tensors - there is no CPU path")
        if self.training and torch.is_grad_enabled():
            raise _lib.EEMFlowHipError("EEMFlow_cdc.forward: inference only (call under torch.no_grad() / eval())")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        e1, e2 = events1.contiguous().float(), events2.contiguous().float()
        if e1.shape != e2.shape or e1.dim() != 4 or e1.shape[1] != self.n_first_channels:
            raise ValueError(f"expected two (B,{self.n_first_channels},H,W) tensors")
        b, _, h, w = e1.shape
        ctx = self._context(e1.device)
        out = torch.empty(5, b, 2, h, w, device=e1.device, dtype=torch.float32)
        padc = (ctypes.c_int * 4)(*self.image_padder._pad)
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().eemplus_forward(ctx, e1.data_ptr(), e2.data_ptr(), b, h, w, ctypes.byref(padc), out.data_ptr(),
                                                  _lib.current_stream_ptr(e1.device)))
        return (events1, events2), [out[i] for i in range(5)]

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
