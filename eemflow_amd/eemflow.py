"""EEMFlow with the reference's nn.Module interface, computed by libeemflow_hip.so on MI355X.

Drop-in for `model.EEMFlow.EEMFlow.EEMFlow` (reference: model/EEMFlow/EEMFlow.py:71-183):
same constructor, `change_imagesize`, `forward(events1, events2) -> ((events1, events2), [flow])`,
`upsample_flow`, and the same 66-tensor state_dict, so test_EEMFlow_HREM.py / train_mvsec.py's
run_network call it unchanged.  The parameters are ordinary nn.Parameters (checkpoint layout and
optimizers keep working); the forward hands them, flattened, to the HIP library, which packs them
into MFMA fragment order once per weight version.

Two routes through the library, chosen as nn.Module semantics dictate:
* no gradient needed (torch.no_grad() / no parameter requires grad): `eemflow_forward`, the HIP-graph replay;
* gradient needed: `_EEMFlowFunction`, a torch.autograd.Function over `eemflow_forward_train` / `eemflow_backward`,
  so the reference trainer's own sequence (train_mvsec.py:245-258: model(im1, im2) -> sequence_loss ->
  scaler.scale(loss).backward() -> clip_grad_norm_ -> optimizer.step()) fills nn.Parameter.grad and works with any
  torch optimizer / loss.  After an optimizer step the changed parameters reach the device copy by one
  device-to-device gather (`eemflow_update_weights`), detected through the parameters' version counters.
forward() requires CUDA (ROCm) tensors: there is no CPU path.  Writes that bypass the version counter
(`p.data.copy_`, `torch.distributed.broadcast(p.data)`) need `model.invalidate_weights()`.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .padder import InputPadder
from .weights import CORR_TAPS_53, eemflow_param_shapes


def convrelu(in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True):
    # parameter container only (keys '<name>.0.weight' / '<name>.0.bias' as in EEMFlow.py:26-30)
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias=bias),
        nn.LeakyReLU(0.1, inplace=True))


class Decoder(nn.Module):
    """Parameter container mirroring EEMFlow.py:37-46."""

    def __init__(self, in_channels, groups):
        super().__init__()
        self.in_channels = in_channels
        self.groups = groups
        self.conv1 = convrelu(in_channels, 100, 3, 1)
        self.conv2 = convrelu(100, 100, 3, 1, groups=groups)
        self.conv3 = convrelu(100, 100, 3, 1, groups=groups)
        self.conv4 = convrelu(100, 100, 3, 1, groups=groups)
        self.conv5 = convrelu(100, 64, 3, 1)
        self.conv6 = convrelu(64, 32, 3, 1)
        self.conv7 = nn.Conv2d(32, 2, 3, 1, 1)


class _EEMFlowFunction(torch.autograd.Function):
    """autograd through the HIP forward: d loss / d flow -> d loss / d parameter (autograd of EEMFlow.py:122-183).
    The event volumes get no gradient (the reference never asks for one: they are data)."""

    @staticmethod
    def forward(ctx, module, e1, e2, out_size, *params):
        handle = module._context(e1.device)
        b, _, h, w = e1.shape
        flow = torch.empty(b, 2, out_size[0], out_size[1], device=e1.device, dtype=torch.float32)
        serial = ctypes.c_int64()
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().eemflow_forward_train(handle, e1.data_ptr(), e2.data_ptr(), b, h, w, flow.data_ptr(),
                                                        out_size[0], out_size[1], ctypes.byref(serial),
                                                        _lib.current_stream_ptr(e1.device)))
        ctx.module, ctx.serial, ctx.out_size = module, serial.value, out_size
        ctx.image_size = (int(module.image_size[0]), int(module.image_size[1]))     # the padder this forward ran with
        ctx.weights_version = module._weights_fingerprint()      # the tuple itself: the module's cache field may be reset (invalidate_weights, reload)
        ctx.save_for_backward(e1, e2)
        return flow

    @staticmethod
    def backward(ctx, dflow):
        m = ctx.module
        e1, e2 = ctx.saved_tensors
        L = _lib.lib()
        if m._weights_fingerprint() != ctx.weights_version:
            raise _lib.EEMFlowHipError("EEMFlow backward: a parameter was modified in place between forward and backward")
        dflow = dflow.contiguous().float()
        n = sum(p.numel() for p in m.parameters())
        grad = torch.empty(n, device=e1.device, dtype=torch.float32)
        b, _, h, w = e1.shape
        with torch.cuda.device(e1.device):
            s = _lib.current_stream_ptr(e1.device)
            if L.eemflow_backward(m._ctx, ctx.serial, e1.data_ptr(), e2.data_ptr(), dflow.data_ptr(), grad.data_ptr(), s) != 0:
                # another forward of this module ran in between and reused the workspace: recompute the activations, with the
                # padder of THIS graph's forward (a validation forward may have brought another image size; the module's next
                # forward sets its own again in _context)
                _lib.check(L.eemflow_set_image_size(m._ctx, ctx.image_size[0], ctx.image_size[1], None))
                scratch = torch.empty(b, 2, ctx.out_size[0], ctx.out_size[1], device=e1.device, dtype=torch.float32)
                serial = ctypes.c_int64()
                _lib.check(L.eemflow_forward_train(m._ctx, e1.data_ptr(), e2.data_ptr(), b, h, w, scratch.data_ptr(),
                                                   ctx.out_size[0], ctx.out_size[1], ctypes.byref(serial), s))
                _lib.check(L.eemflow_backward(m._ctx, serial.value, e1.data_ptr(), e2.data_ptr(), dflow.data_ptr(),
                                              grad.data_ptr(), s))
        grads, off = [], 0
        for i, p in enumerate(m.parameters()):
            k = p.numel()
            grads.append(grad[off:off + k].view_as(p) if ctx.needs_input_grad[4 + i] else None)
            off += k
        return (None, None, None, None, *grads)


class EEMFlow(nn.Module):
    def __init__(self, config, groups=5, n_first_channels=5, out_mesh_size=False):
        super().__init__()
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
        # 53-tap diamond (EEMFlow+.py:89-97); the 49-entry list of EEMFlow.py:85-94 cannot feed
        # Decoder(69).  Plain attribute, not a buffer: it is not part of the checkpoint.
        self.index = torch.tensor(CORR_TAPS_53)
        self.rconv_1 = convrelu(16, 16, 3, 1)
        self.rconv_2 = convrelu(32, 16, 3, 1)
        self.rconv_3 = convrelu(64, 16, 3, 1)
        self.decoder_1 = Decoder(69, groups)
        self.decoder_2 = Decoder(69, groups)
        self.decoder_3 = Decoder(69, groups)
        self.out_conv = nn.Conv2d(6, 2, 1, 1)
        self.out_mesh_size = out_mesh_size
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        assert list(self.state_dict().keys()) == list(eemflow_param_shapes(n_first_channels, groups).keys())
        self._ctx = None
        self._ctx_device = None
        self._weights_version = None
        self._layout_loaded = False
        self.use_graph = True
        self.frames_in_flight = 1       # >= 3: this module is one of several replicas kept busy on separate streams (throughput over latency)

    # ------------------------------------------------------------------ reference interface
    def change_imagesize(self, img_size):
        self.image_size = img_size
        self.image_padder = InputPadder(img_size, mode='chairs', eval_pad_rate=64)

    def replicate(self, frames_in_flight=None):
        """A second module with the same weights, device, image size and mode and a context of its own: what keeps one more frame
        in flight on another HIP stream (harness.TestRaftEvents(frames_in_flight=...))."""
        twin = EEMFlow("", groups=self.groups, n_first_channels=self.n_first_channels, out_mesh_size=self.out_mesh_size)
        twin.load_state_dict(self.state_dict())
        twin = twin.to(next(self.parameters()).device)
        if hasattr(self, "image_size"):
            twin.change_imagesize(self.image_size)
        twin.train(self.training)
        twin.frames_in_flight = self.frames_in_flight if frames_in_flight is None else frames_in_flight
        return twin

    def upsample_flow(self, flow, orig_size):
        if not flow.is_cuda:
            raise _lib.EEMFlowHipError("EEMFlow.upsample_flow: the HIP path needs a CUDA (ROCm) tensor")
        flow = flow.contiguous().float()
        b, c, h, w = flow.shape
        out = torch.empty(b, c, int(orig_size[0]), int(orig_size[1]), device=flow.device, dtype=torch.float32)
        with torch.cuda.device(flow.device):
            _lib.check(_lib.lib().eemflow_upsample_bilinear(
                flow.data_ptr(), out.data_ptr(), b * c, h, w, out.shape[2], out.shape[3],
                _lib.current_stream_ptr(flow.device)))
        return out

    def forward(self, events1, events2):
        if not (events1.is_cuda and events2.is_cuda):
            raise _lib.EEMFlowHipError(
                "EEMFlow.forward: inputs must be CUDA (ROCm) tensors - this implementation has no CPU path")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        input_size = events1.shape[-2:]
        if self.training and self.out_mesh_size:
            out_size = (16, 16)                                  # EEMFlow.py:126-130
        else:
            out_size = tuple(int(v) for v in input_size)
        e1 = events1.contiguous().float()
        e2 = events2.contiguous().float()
        if e1.shape != e2.shape or e1.dim() != 4 or e1.shape[1] != self.n_first_channels:
            raise ValueError(f"expected two (B,{self.n_first_channels},H,W) tensors, got {tuple(e1.shape)} and {tuple(e2.shape)}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            flow = _EEMFlowFunction.apply(self, e1, e2, out_size, *self.parameters())
            return (events1, events2), [flow]
        ctx = self._context(e1.device)
        b, _, h, w = e1.shape
        flow = torch.empty(b, 2, out_size[0], out_size[1], device=e1.device, dtype=torch.float32)
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().eemflow_forward(ctx, e1.data_ptr(), e2.data_ptr(), b, h, w, flow.data_ptr(),
                                                  out_size[0], out_size[1], _lib.current_stream_ptr(e1.device)))
        return (events1, events2), [flow]

    MAX_COALESCE = 16

    def forward_many(self, frames, deferred_norm=False):
        """Several INDEPENDENT samples of the evaluation loop (test_mvsec.py:580-597: one `model(events1, events2)` per sample at batch 1)
        as one batch-n chain of launches, each frame staying in its own tensors: `frames` is a sequence of (events1, events2) pairs of
        [1, C, H, W] tensors; returns one `((events1, events2), [flow])` per frame - flow [1, 2, H, W], bitwise what `forward` gives for
        the frames stacked into one batch.  Inference only (no autograd graph is recorded).
        deferred_norm=True: the frames are RAW voxel grids with their normalisation record behind them (the voxelizer's
        normalize="deferred"); pconv1_1 applies loader_utils.py:527-535's (v - mean) / sd as it reads them."""
        frames = list(frames)
        if not 1 <= len(frames) <= self.MAX_COALESCE:
            raise ValueError(f"forward_many: 1..{self.MAX_COALESCE} frames per call, got {len(frames)}")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        keep, shape = [], None
        for a, b in frames:
            if not (a.is_cuda and b.is_cuda):
                raise _lib.EEMFlowHipError("EEMFlow.forward_many: inputs must be CUDA (ROCm) tensors - this implementation has no CPU path")
            a, b = a.contiguous().float(), b.contiguous().float()
            if a.shape != b.shape or a.dim() != 4 or a.shape[0] != 1 or a.shape[1] != self.n_first_channels:
                raise ValueError(f"forward_many: every frame is two (1,{self.n_first_channels},H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
            if shape is not None and a.shape != shape:
                raise ValueError("forward_many: all frames of a call share one shape")
            if deferred_norm:
                from .voxelizer import has_norm_record
                if not (has_norm_record(a) and has_norm_record(b)):
                    raise ValueError("forward_many(deferred_norm=True): every volume needs its four-float record behind it "
                                     "(voxelize with normalize='deferred')")
            shape = a.shape
            keep.append((a, b))
        dev = keep[0][0].device
        h, w = int(shape[2]), int(shape[3])
        out_size = (16, 16) if (self.training and self.out_mesh_size) else (h, w)
        ctx = self._context(dev)
        _lib.check(_lib.lib().eemflow_set_deferred_input_norm(ctx, 1 if deferred_norm else 0))
        n = len(keep)
        flows = [torch.empty(1, 2, out_size[0], out_size[1], device=dev, dtype=torch.float32) for _ in range(n)]
        arr = ctypes.c_void_p * n
        p1, p2, po = arr(*[a.data_ptr() for a, _ in keep]), arr(*[b.data_ptr() for _, b in keep]), arr(*[f.data_ptr() for f in flows])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().eemflow_forward_many(ctx, n, p1, p2, po, h, w, out_size[0], out_size[1], _lib.current_stream_ptr(dev)))
        return [((frames[i][0], frames[i][1]), [flows[i]]) for i in range(n)]

    # ------------------------------------------------------------------ HIP context plumbing
    def _flat_weights(self, device=None):
        # state_dict order == parameter registration order (the module has no buffers)
        return torch.cat([v.detach().reshape(-1).to(torch.float32) for v in self.state_dict().values()]).to(device or "cpu")

    def invalidate_weights(self):
        """Force the next forward to re-read the nn.Parameters (after writes through `.data`, which bypass the
        version counter the staleness check reads)."""
        self._weights_version = None

    def _weights_fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _context(self, device):
        L = _lib.lib()
        if self._ctx is None or self._ctx_device != device:
            self._release()
            handle = ctypes.c_void_p()
            _lib.check(L.eemflow_create(device.index if device.index is not None else torch.cuda.current_device(),
                                        ctypes.byref(handle)))
            self._ctx, self._ctx_device, self._weights_version, self._layout_loaded = handle, device, None, False
        fp = self._weights_fingerprint()
        if fp != self._weights_version:
            if self._layout_loaded and all(p.device == device for p in self.parameters()):
                flat = self._flat_weights(device).contiguous()          # e.g. after optimizer.step(): device to device
                with torch.cuda.device(device):
                    _lib.check(L.eemflow_update_weights(self._ctx, flat.data_ptr(), flat.numel(), _lib.current_stream_ptr(device)))
            else:
                flat = self._flat_weights().contiguous()
                _lib.check(L.eemflow_load_weights(self._ctx, flat.data_ptr(), flat.numel(), self.n_first_channels, self.groups))
                self._layout_loaded = True
            self._weights_version = fp
        pad = (ctypes.c_int * 4)()
        _lib.check(L.eemflow_set_image_size(self._ctx, int(self.image_size[0]), int(self.image_size[1]), ctypes.byref(pad)))
        assert list(pad) == self.image_padder._pad
        _lib.check(L.eemflow_use_graph(self._ctx, 1 if self.use_graph else 0))
        _lib.check(L.eemflow_set_frames_in_flight(self._ctx, max(1, int(self.frames_in_flight))))
        _lib.check(L.eemflow_set_deferred_input_norm(self._ctx, 0))       # (forward_many(deferred_norm=True) turns it on for its call)
        return self._ctx

    def stage(self, name):
        """Intermediate tensor of the last forward (parity tests): see eemflow_get_stage."""
        L = _lib.lib()
        dims = (ctypes.c_int * 4)()
        _lib.check(L.eemflow_get_stage(self._ctx, name.encode(), None, 0, ctypes.byref(dims), None))
        out = torch.empty(*list(dims), device=self._ctx_device, dtype=torch.float32)
        with torch.cuda.device(self._ctx_device):
            _lib.check(L.eemflow_get_stage(self._ctx, name.encode(), out.data_ptr(), out.numel(), ctypes.byref(dims),
                                           _lib.current_stream_ptr(self._ctx_device)))
        return out

    def _release(self):
        if self._ctx is not None:
            _lib.lib().eemflow_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
