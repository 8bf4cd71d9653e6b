"""E-RAFT on MI355X behind the reference's nn.Module interfaces.

Mirrors model/eraft.py (ERAFT), model/extractor.py (BasicEncoder, ResidualBlock), model/update.py
(BasicUpdateBlock and parts) and model/corr.py (CorrBlock): same constructors, forward signatures and
state_dict keys (179 tensors).  The modules only hold parameters; all arithmetic happens in
libeemflow_hip.so (include/eemflow_hip.h).  CUDA tensors only.  Two routes, as nn.Module semantics dictate:
* no gradient needed: `eraft_forward`, the fused inference schedule (eval-mode BatchNorm folded into the convs);
* gradient needed (train_mvsec.py:245-258 on this model): the forward is assembled from the operator-level
  autograd Functions of eemflow_amd/ops.py (eemop_* entry points): train-mode BatchNorm with batch statistics and
  running-statistics update in cnet, InstanceNorm in fnet, the 12 unrolled update iterations (SepConvGRU, motion encoder,
  flow / mask heads) through shared weights, the pyramid lookup at detached coordinates, convex upsampling.
"""
import ctypes
from argparse import Namespace

import torch
import torch.nn as nn

from . import _lib
from .padder import InputPadder


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        if norm_fn == 'batch':
            mk = lambda: nn.BatchNorm2d(planes)
        elif norm_fn == 'instance':
            mk = lambda: nn.InstanceNorm2d(planes)
        elif norm_fn == 'none':
            mk = lambda: nn.Sequential()                         # model/extractor.py:36-40
        elif norm_fn == 'group':
            mk = lambda: nn.GroupNorm(num_groups=planes // 8, num_channels=planes)      # model/extractor.py:19-23
        else:
            raise ValueError(f"norm_fn {norm_fn!r}: 'group', 'batch', 'instance' or 'none' (model/extractor.py:17-40)")
        self.norm1, self.norm2 = mk(), mk()
        if not stride == 1:
            self.norm3 = mk()
        if stride == 1:
            self.downsample = None
        else:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)
        self.stride = stride


class BasicEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0, n_first_channels=1):
        super().__init__()
        self.norm_fn = norm_fn
        if norm_fn == 'batch':
            self.norm1 = nn.BatchNorm2d(64)
        elif norm_fn == 'instance':
            self.norm1 = nn.InstanceNorm2d(64)
        elif norm_fn == 'none':
            self.norm1 = nn.Sequential()                         # model/extractor.py:131-132
        elif norm_fn == 'group':
            self.norm1 = nn.GroupNorm(num_groups=8, num_channels=64)                     # model/extractor.py:123-124
        else:
            raise ValueError(f"norm_fn {norm_fn!r}: 'group', 'batch', 'instance' or 'none' (model/extractor.py:122-132)")
        self.conv1 = nn.Conv2d(n_first_channels, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=2)
        self.layer3 = self._make_layer(128, stride=2)
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None                # model/extractor.py:147-149
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                  ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x):
        """model/extractor.py:162-190 on the HIP operators (differentiable): a list / tuple of inputs runs as one batch and comes back
        split (:165-169,186-188).  ERAFT calls the same operator chain (ERAFT._encoder_ops) with its own batching."""
        from . import ops
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        if not x.is_cuda:
            raise _lib.EEMFlowHipError("BasicEncoder.forward: CUDA (ROCm) tensors required - there is no CPU path")
        y = ops.conv2d(self.conv2, encoder_ops(self, x.contiguous().float()))
        if self.training and self.dropout is not None:                     # model/extractor.py:183-184
            y = ops.dropout2d(y, self.dropout.p)
        if is_list:
            y = torch.split(y, [batch_dim, batch_dim], dim=0)
        return y


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))


class BasicMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


class BasicUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim=128, input_dim=128):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(
            nn.Conv2d(hidden_dim, hidden_dim * 2, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(hidden_dim * 2, 64 * 9, 1, padding=0))


def get_args():
    return Namespace(small=False, dropout=False, mixed_precision=False, clip=1.0)


class ERAFT(nn.Module):
    def __init__(self, config, n_first_channels=5):
        super().__init__()
        args = get_args()
        self.args = args
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        self.n_first_channels = n_first_channels
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=0, n_first_channels=n_first_channels)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn='batch', dropout=0, n_first_channels=n_first_channels)
        self.update_block = BasicUpdateBlock(self.args, hidden_dim=hdim)
        self._ctx = None
        self._ctx_device = None
        self._weights_version = None
        self.keep_stages = False       # True: the first iteration's corr0 / net1 / mask1 / delta1 stay readable through stage()
        self.frames_in_flight = 1      # >= 3: one of several replicas kept busy on separate streams (eraft_set_frames_in_flight)
        self.final_only = False        # True (inference route): the returned list holds the last prediction only - what test_mvsec.py:1455 reads
        self.alternate_corr = False    # True (inference route): correlation features on the fly, no all-pairs volume (RAFT's alternate_corr)

    def change_imagesize(self, img_size):
        self.image_size = img_size
        self.image_padder = InputPadder(img_size, mode='chairs')

    def replicate(self, frames_in_flight=None):
        """A second module with the same weights, device, image size and mode and a context of its own: what keeps one more frame
        in flight on another HIP stream (harness.TestRaftEvents(frames_in_flight=...))."""
        twin = ERAFT("", n_first_channels=self.n_first_channels)
        twin.load_state_dict(self.state_dict())
        twin = twin.to(next(self.parameters()).device)
        if hasattr(self, "image_size"):
            twin.change_imagesize(self.image_size)
        twin.train(self.training)
        twin.frames_in_flight = self.frames_in_flight if frames_in_flight is None else frames_in_flight
        twin.final_only = getattr(self, "final_only", False)
        twin.alternate_corr = getattr(self, "alternate_corr", False)
        return twin

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    # ------------------------------------------------------------------ HIP plumbing
    def _flat_weights(self):
        """All float tensors of the state_dict in registration order (num_batches_tracked skipped)."""
        return torch.cat([v.detach().reshape(-1).to(torch.float32).cpu()
                          for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")])

    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _context(self, device):
        L = _lib.lib()
        if self._ctx is None or self._ctx_device != device:
            self._release()
            handle = ctypes.c_void_p()
            _lib.check(L.eraft_create(device.index if device.index is not None else torch.cuda.current_device(),
                                      ctypes.byref(handle)))
            self._ctx, self._ctx_device, self._weights_version = handle, device, None
        fp = self._fingerprint()
        if fp != self._weights_version:
            flat = self._flat_weights().contiguous()
            _lib.check(L.eraft_load_weights(self._ctx, flat.data_ptr(), flat.numel(), self.n_first_channels))
            self._weights_version = fp
        _lib.check(L.eraft_keep_stages(self._ctx, 1 if self.keep_stages else 0))
        _lib.check(L.eraft_set_frames_in_flight(self._ctx, max(1, int(getattr(self, "frames_in_flight", 1)))))
        _lib.check(L.eraft_set_alternate_corr(self._ctx, 1 if getattr(self, "alternate_corr", False) else 0))
        _lib.check(L.eraft_set_final_only(self._ctx, 1 if getattr(self, "final_only", False) else 0))
        return self._ctx

    def forward(self, events1, events2, iters=12, flow_init=None, upsample=True, normal=False):
        if not (events1.is_cuda and events2.is_cuda):
            raise _lib.EEMFlowHipError("ERAFT.forward: inputs must be CUDA (ROCm) tensors - there is no CPU path")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        e1, e2 = events1.contiguous().float(), events2.contiguous().float()
        if e1.shape != e2.shape or e1.dim() != 4 or e1.shape[1] != self.n_first_channels:
            raise ValueError(f"expected two (B,{self.n_first_channels},H,W) tensors")
        b, _, h, w = e1.shape
        pad = self.image_padder._pad
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        if hp % 8 or wp % 8:
            raise ValueError(f"padded size {hp}x{wp} is not a multiple of 8: the reference's convex upsampling "
                             "(eraft.py:83-94) cannot be unpadded consistently")
        bn_train = any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d))
        if (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())) or bn_train:
            # autograd route; also taken without gradients while BatchNorm is in train(): batch statistics, running-stat update
            return (events1, events2), self._forward_ops(e1, e2, iters, flow_init)
        ctx = self._context(e1.device)
        nout = 1 if getattr(self, "final_only", False) else iters
        out = torch.empty(nout, b, 2, h, w, device=e1.device, dtype=torch.float32)
        fi = None
        if flow_init is not None:
            fi = flow_init.contiguous().float()
        padc = (ctypes.c_int * 4)(*pad)
        with torch.cuda.device(e1.device):
            _lib.check(_lib.lib().eraft_forward(ctx, e1.data_ptr(), e2.data_ptr(), b, h, w, ctypes.byref(padc), iters,
                                                fi.data_ptr() if fi is not None else None, out.data_ptr(),
                                                _lib.current_stream_ptr(e1.device)))
        return (events1, events2), [out[i] for i in range(nout)]

    MAX_COALESCE = 16

    def forward_many(self, frames, iters=12):
        """Several INDEPENDENT samples of the evaluation loop (test_mvsec.py:580-597: one `model(events1, events2)` per sample at batch 1)
        as one batch-n forward, each sample staying in its own tensors: `frames` is a sequence of (events1, events2) pairs of [1, C, H, W]
        tensors; returns one `((events1, events2), [predictions])` per sample - `iters` predictions, or the last one with `final_only` -
        bitwise what `forward` gives for the samples stacked into one batch.  Inference only, no flow_init."""
        frames = list(frames)
        if not 1 <= len(frames) <= self.MAX_COALESCE:
            raise ValueError(f"forward_many: 1..{self.MAX_COALESCE} frames per call, got {len(frames)}")
        if not hasattr(self, "image_padder"):
            raise AttributeError("call change_imagesize(img_size) before forward (as the reference requires)")
        keep, shape = [], None
        for a, b in frames:
            if not (a.is_cuda and b.is_cuda):
                raise _lib.EEMFlowHipError("ERAFT.forward_many: inputs must be CUDA (ROCm) tensors - there is no CPU path")
            a, b = a.contiguous().float(), b.contiguous().float()
            if a.shape != b.shape or a.dim() != 4 or a.shape[0] != 1 or a.shape[1] != self.n_first_channels:
                raise ValueError(f"forward_many: every frame is two (1,{self.n_first_channels},H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
            if shape is not None and a.shape != shape:
                raise ValueError("forward_many: all frames of a call share one shape")
            shape = a.shape
            keep.append((a, b))
        dev = keep[0][0].device
        h, w = int(shape[2]), int(shape[3])
        pad = self.image_padder._pad
        if (h + pad[2] + pad[3]) % 8 or (w + pad[0] + pad[1]) % 8:
            raise ValueError("forward_many: the padded size must be a multiple of 8 (eraft.py:83-94)")
        ctx = self._context(dev)
        n = len(keep)
        nout = 1 if getattr(self, "final_only", False) else iters
        outs = [torch.empty(nout, 1, 2, h, w, device=dev, dtype=torch.float32) for _ in range(n)]
        arr = ctypes.c_void_p * n
        p1, p2, po = arr(*[a.data_ptr() for a, _ in keep]), arr(*[b.data_ptr() for _, b in keep]), arr(*[o.data_ptr() for o in outs])
        padc = (ctypes.c_int * 4)(*pad)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().eraft_forward_many(ctx, n, p1, p2, h, w, padc, iters, po, _lib.current_stream_ptr(dev)))
        return [((frames[i][0], frames[i][1]), [outs[i][k] for k in range(nout)]) for i in range(n)]

    # ------------------------------------------------------------------ differentiable route (eemflow_amd/ops.py)
    def _norm(self, norm, x, relu):
        return apply_norm(norm, x, relu)

    def _encoder_ops(self, enc, x):
        return encoder_ops(enc, x)

    def _update_ops(self, net, inp, corr, flow):
        """BasicUpdateBlock.forward (model/update.py:97-106)."""
        from . import ops
        ub = self.update_block
        e = ub.encoder
        cor = ops.conv2d(e.convc2, ops.conv2d(e.convc1, corr, act=ops.ACT_RELU), act=ops.ACT_RELU)
        flo = ops.conv2d(e.convf2, ops.conv2d(e.convf1, flow, act=ops.ACT_RELU), act=ops.ACT_RELU)
        motion = ops.Cat2.apply(ops.conv2d(e.conv, cor, flo, act=ops.ACT_RELU), flow)
        # every tensor with several consumers goes through ops.fan_out: its gradients meet in one launch instead of a chain of adds
        mo = ops.fan_out(motion, 6)
        ip = ops.fan_out(inp, 6)
        g = ub.gru
        for k, (cz, cr, cq) in enumerate(((g.convz1, g.convr1, g.convq1), (g.convz2, g.convr2, g.convq2))):
            nz, nr, nm, nb = ops.fan_out(net, 4)
            z = ops.conv2d(cz, nz, ip[3 * k], mo[3 * k], act=ops.ACT_SIGMOID)
            r = ops.conv2d(cr, nr, ip[3 * k + 1], mo[3 * k + 1], act=ops.ACT_SIGMOID)
            q = ops.conv2d(cq, ops.Mul.apply(r, nm), ip[3 * k + 2], mo[3 * k + 2], act=ops.ACT_TANH)
            net = ops.GRUBlend.apply(z, nb, q)
        nf, nk, net = ops.fan_out(net, 3)
        delta = ops.conv2d(ub.flow_head.conv2, ops.conv2d(ub.flow_head.conv1, nf, act=ops.ACT_RELU))
        mask = ops.conv2d(ub.mask[2], ops.conv2d(ub.mask[0], nk, act=ops.ACT_RELU), out_scale=0.25)
        return net, mask, delta

    def _forward_ops(self, e1, e2, iters, flow_init):
        """ERAFT.forward (model/eraft.py:97-159) as an autograd graph of HIP operators."""
        from . import ops
        if not (e1.is_cuda and all(p.is_cuda for p in self.parameters())):
            raise _lib.EEMFlowHipError("ERAFT.forward: CUDA (ROCm) tensors required - there is no CPU path")
        b, c, h, w = e1.shape
        pad = self.image_padder._pad
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        x = torch.empty(2 * b, c, hp, wp, device=e1.device, dtype=torch.float32)
        ops.replicate_pad_into(e1, pad, x[:b])
        ops.replicate_pad_into(e2, pad, x[b:])
        fmap = ops.conv2d(self.fnet.conv2, self._encoder_ops(self.fnet, x))               # [image1; image2] as one batch (:116)
        pyr = ops.CorrPyramid.apply(fmap[:b], fmap[b:])
        ctx_feat = self._encoder_ops(self.cnet, x[:b])
        w2, b2 = self.cnet.conv2.weight, self.cnet.conv2.bias
        hd = self.hidden_dim
        cfa, cfb = ops.fan_out(ctx_feat, 2)
        net = ops.conv2d(self.cnet.conv2, cfa, act=ops.ACT_TANH, weight=w2[:hd], bias=b2[:hd])      # :128-131
        inp = ops.conv2d(self.cnet.conv2, cfb, act=ops.ACT_RELU, weight=w2[hd:], bias=b2[hd:])
        coords0, coords1 = ops.coords_grids(b, hp // 8, wp // 8, e1.device, flow_init)
        preds = []
        inps = ops.fan_out(inp, iters)                                                     # (one alias per iteration; _update_ops fans each out again)
        pyrs = [ops.fan_out(p, iters) for p in pyr]
        for it in range(iters):
            coords1 = coords1.detach()                                                     # :141
            corr = ops.CorrLookup.apply(coords1, *[p[it] for p in pyrs])
            net, mask, delta = self._update_ops(net, inps[it], corr, ops.sub(coords1, coords0))
            coords1 = ops.Add.apply(coords1, delta, 1)
            preds.append(ops.ConvexUpsample.apply(ops.Add.apply(coords1, coords0, -1), mask, tuple(pad)))
        return preds

    def stage(self, name):
        L = _lib.lib()
        dims = (ctypes.c_int * 4)()
        _lib.check(L.eraft_get_stage(self._ctx, name.encode(), None, 0, ctypes.byref(dims), None))
        out = torch.empty(*list(dims), device=self._ctx_device, dtype=torch.float32)
        with torch.cuda.device(self._ctx_device):
            _lib.check(L.eraft_get_stage(self._ctx, name.encode(), out.data_ptr(), out.numel(), ctypes.byref(dims),
                                         _lib.current_stream_ptr(self._ctx_device)))
        return out

    def _release(self):
        if self._ctx is not None:
            _lib.lib().eraft_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


# ---------------------------------------------------------------------- BasicEncoder's operator chain (module level: BasicEncoder.forward and ERAFT share it)
def apply_norm(norm, x, relu):
    """norm + optional ReLU as one HIP operator."""
    from . import ops
    if isinstance(norm, nn.InstanceNorm2d):
        return ops.InstanceNormReLU.apply(x, relu)
    if isinstance(norm, nn.GroupNorm):
        return ops.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps, relu)
    if not norm.training:                                        # freeze_bn() (model/eraft.py:69-72): running statistics, no update
        return ops.BatchNormEvalReLU.apply(x, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps, relu)
    y = ops.BatchNormTrainReLU.apply(x, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.momentum, norm.eps, relu)
    norm.num_batches_tracked += 1
    return y


def conv_norm(conv, norm, x, relu):
    """conv -> norm -> [ReLU]; norm_fn 'none' is the reference's empty nn.Sequential (model/extractor.py:36-40): the ReLU then rides in the
    convolution's epilogue."""
    from . import ops
    if isinstance(norm, nn.Sequential) and len(norm) == 0:
        return ops.conv2d(conv, x, act=ops.ACT_RELU if relu else ops.ACT_NONE)
    return apply_norm(norm, ops.conv2d(conv, x), relu)


def encoder_ops(enc, x):
    """BasicEncoder up to (not including) its 1x1 output conv (model/extractor.py:170-185)."""
    from . import ops
    y = conv_norm(enc.conv1, enc.norm1, x, True)
    for layer in (enc.layer1, enc.layer2, enc.layer3):
        for blk in layer:                                                  # ResidualBlock.forward, model/extractor.py:43-57
            ya, yb = ops.fan_out(y, 2)                                     # (the block's input feeds the convs and the shortcut)
            t = conv_norm(blk.conv1, blk.norm1, ya, True)
            t = conv_norm(blk.conv2, blk.norm2, t, True)
            if blk.downsample is not None:
                yb = conv_norm(blk.downsample[0], blk.norm3, yb, False)
            y = ops.AddReLU.apply(yb, t)
    return y
