"""torch.autograd.Functions over the operator-level C ABI (eemop_* / eraft_*_bwd in include/eemflow_hip.h).

These are what `ERAFT.forward` is made of when the caller needs gradients (train_mvsec.py:245-258 on model/eraft.py): autograd keeps
the graph - the 12 unrolled update iterations that share one set of weights, the detached coordinates, the accumulation of weight
gradients - and every node's arithmetic, forward and backward, runs in libeemflow_hip.so.  torch supplies tensors (allocation, views,
slicing of gradients), nothing else: no ATen compute op is on the path.  CUDA (ROCm) tensors only; there is no CPU path.
"""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3


def _sp(t):
    return _lib.current_stream_ptr(t.device)


_NULL_CTX = contextlib.nullcontext()


def _on(device):
    """`with torch.cuda.device(device)` only when it is not the current one already (the context manager costs ~10 us per operator,
    a training step of E-RAFT runs ~3 000 of them and is host-bound)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NULL_CTX
    return torch.cuda.device(device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EEMFlowHipError("eemflow_amd.ops: CUDA (ROCm) tensors required - there is no CPU path")


def _act_bwd(dy, y, kind, scale=1.0):
    out = torch.empty_like(y)
    with _on(y.device):
        _lib.check(_lib.lib().eemop_act_bwd(_c(dy).data_ptr(), y.data_ptr(), y.numel(), kind, float(scale), out.data_ptr(), _sp(y)))
    return out


def _binary(kind, a, b=None, alpha=1.0):
    out = torch.empty_like(a)
    with _on(a.device):
        _lib.check(_lib.lib().eemop_binary(kind, a.data_ptr(), _ptr(b), float(alpha), a.numel(), out.data_ptr(), _sp(a)))
    return out


_side_streams = {}


def _wgrad_stream(device):
    """Side stream of the weight-gradient launches (a leaf of the backward chain: it runs beside the data gradient of the same layer
    and is joined before backward() returns); None with EEM_NO_WGRAD_STREAM=1."""
    if os.environ.get("EEM_NO_WGRAD_STREAM", "0") == "1":
        return None
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# ---- packed-weight cache (include/eemflow_hip.h: eemop_pack_hint): a token per weight tensor that is never reused + its version counter
_pack_tokens = {}                                               # id(parameter) -> (token, weak reference): tensors do not hash by value
_next_token = [1]


def _forget_packs(key, token):
    ent = _pack_tokens.get(key)
    if ent is not None and ent[0] == token:
        del _pack_tokens[key]
    try:
        _lib.lib().eemop_pack_forget(token)
    except Exception:                                           # interpreter shutdown: the library may be gone
        pass


def _pack_identity(w):
    """(token, version) of the tensor whose storage `w` reads: nn.Parameters (and views of them: the cnet's split output conv) keep
    their packed weights between calls; anything else (a temporary) is packed per call.  In-place updates bump the version; writes
    through `.data` do not - call `invalidate_packed_weights()` after those."""
    base = w._base if w._base is not None else w
    if not isinstance(base, torch.nn.Parameter) or os.environ.get("EEM_NO_PACK_CACHE", "0") == "1":
        return 0, -1
    key = id(base)
    ent = _pack_tokens.get(key)
    if ent is None or ent[1]() is not base:                     # (an id can come back after its tensor died)
        tok = _next_token[0]
        _next_token[0] += 1
        _pack_tokens[key] = ent = (tok, weakref.ref(base))
        weakref.finalize(base, _forget_packs, key, tok)
    return ent[0], base._version + _pack_epoch[0]


_pack_epoch = [0]


def invalidate_packed_weights():
    """Forces every cached packing to be redone at its next use (after weight writes that bypass the version counter, e.g. `p.data`)."""
    _pack_epoch[0] += 1 << 32


class _packs_of:
    """with _packs_of(token, version): the conv launches inside name their weight tensor to the library."""

    def __init__(self, ident):
        self.ident = ident

    def __enter__(self):
        if self.ident[0]:
            _lib.lib().eemop_pack_hint(self.ident[0], self.ident[1])

    def __exit__(self, *exc):
        if self.ident[0]:
            _lib.lib().eemop_pack_hint(0, -1)
        return False


# Weight / bias gradients of convs whose weight IS an nn.Parameter (a leaf, not a view) are accumulated by the library straight into
# `parameter.grad` (allocated zeroed on first use) and the Function returns None for them: E-RAFT's update block uses every weight in
# each of its 12 iterations, and autograd's own accumulation was one at::add_ launch and one zero fill per use and parameter (627 +
# 520 launches of a training step).  What is lost: tensor hooks on those parameters and `torch.autograd.grad(..., inputs=[weight])`
# (the gradient lands in .grad instead; a parameter WITH a registered hook is detected and keeps the autograd route) - switch it off for such callers: `ops.set_direct_param_grads(False)` / EEM_NO_DIRECT_WGRAD=1.
_direct_param_grads = [os.environ.get("EEM_NO_DIRECT_WGRAD", "0") != "1"]


def set_direct_param_grads(enabled):
    """Whether conv weight / bias gradients go straight into `.grad` (default) or through autograd's accumulation; returns the old value."""
    old = _direct_param_grads[0]
    _direct_param_grads[0] = bool(enabled)
    return old


def _has_grad_hooks(t):
    """Tensor hooks (`register_hook`: DDP's reducer is one) or post-accumulate hooks on a parameter: they fire from autograd's
    AccumulateGrad node, which the direct route bypasses - such a parameter keeps the autograd route."""
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


def _leaf_param(t):
    return t if (isinstance(t, torch.nn.Parameter) and t.is_leaf and t._base is None and t.requires_grad and t.is_contiguous()
                 and t.dtype == torch.float32 and not _has_grad_hooks(t)) else None


class Conv2d(torch.autograd.Function):
    """out_scale * act(conv2d(cat(xs, 1), w) + b) - nn.Conv2d (+ the activation behind it) of model/extractor.py / model/update.py;
    the inputs' torch.cat is never materialised."""

    @staticmethod
    def forward(ctx, w, b, stride, padding, act, out_scale, *xs):
        _need_cuda(w, *xs)
        xs = [_c(x) for x in xs]
        # the parameters themselves, for the direct gradient accumulation of backward (None for views / temporaries / frozen ones)
        ctx.w_param = _leaf_param(w) if _direct_param_grads[0] else None
        ctx.b_param = _leaf_param(b) if (_direct_param_grads[0] and b is not None) else None
        w = _c(w)
        n, _, hin, win = xs[0].shape
        cout, cin, kh, kw = w.shape
        cs = [x.shape[1] for x in xs]
        assert sum(cs) == cin and 1 <= len(xs) <= 3
        ph, pw = padding
        hout, wout = (hin + 2 * ph - kh) // stride + 1, (win + 2 * pw - kw) // stride + 1
        out = torch.empty(n, cout, hout, wout, device=w.device, dtype=torch.float32)
        px = [x.data_ptr() for x in xs] + [None] * (3 - len(xs))
        pc = cs + [0] * (3 - len(xs))
        ident = _pack_identity(w)
        with _on(w.device), _packs_of(ident):
            _lib.check(_lib.lib().eemop_conv2d_fwd(px[0], pc[0], px[1], pc[1], px[2], pc[2], w.data_ptr(), _ptr(b), n, hin, win, cout, kh, kw,
                                                   stride, ph, pw, act, float(out_scale), out.data_ptr(), cout, 0, _sp(w)))
        ctx.save_for_backward(w, out if act != ACT_NONE else None, *xs)
        ctx.cfg = (stride, ph, pw, act, float(out_scale), b is not None, cs)
        ctx.pack_ident = ident
        return out

    @staticmethod
    def backward(ctx, dout):
        w, out, *xs = ctx.saved_tensors
        stride, ph, pw, act, out_scale, has_b, cs = ctx.cfg
        L = _lib.lib()
        n, _, hin, win = xs[0].shape
        cout, cin, kh, kw = w.shape
        dout = _c(dout)
        if act != ACT_NONE:
            dpre = _act_bwd(dout, out if out_scale == 1.0 else _binary(4, out, alpha=1.0 / out_scale), act, out_scale)
        elif out_scale != 1.0:
            dpre = _binary(4, dout, alpha=out_scale)
        else:
            dpre = dout
        need = ctx.needs_input_grad
        dw = db = None
        dxs = [None] * len(xs)
        with _on(w.device):
            s = _sp(w)
            joined = None
            ret_dw = ret_db = True
            if need[0] or (has_b and need[1]):
                if need[0] and ctx.w_param is not None:                   # accumulate into parameter.grad, hand autograd nothing
                    if ctx.w_param.grad is None:
                        ctx.w_param.grad = torch.zeros_like(ctx.w_param)
                    dw, ret_dw = ctx.w_param.grad, False
                else:
                    dw = torch.zeros_like(w)
                if has_b and need[1] and ctx.b_param is not None:
                    if ctx.b_param.grad is None:
                        ctx.b_param.grad = torch.zeros_like(ctx.b_param)
                    db, ret_db = ctx.b_param.grad, False
                else:
                    db = torch.zeros(cout, device=w.device) if has_b else None
                side = _wgrad_stream(w.device) if any(need[6:]) else None
                sw = s
                if side is not None:                                       # fork: dpre and the zeroed buffers are complete
                    cur = torch.cuda.current_stream(w.device)
                    fork = torch.cuda.Event()
                    fork.record(cur)
                    side.wait_event(fork)
                    sw = ctypes.c_void_p(side.cuda_stream)
                # one call for all input segments (the GRU's [h | inp | motion]): the library takes them in one launch where that is faster
                px = [x.data_ptr() for x in xs] + [None] * (3 - len(xs))
                pc = cs + [0] * (3 - len(xs))
                _lib.check(L.eemop_conv2d_bwd_weight_cat(px[0], pc[0], px[1], pc[1], px[2], pc[2], dpre.data_ptr(), n, hin, win, cout, kh, kw,
                                                         stride, ph, pw, dw.data_ptr(), _ptr(db), sw))
                if side is not None:
                    joined = torch.cuda.Event()
                    joined.record(side)
            c0 = 0
            for i, x in enumerate(xs):
                if need[6 + i]:
                    dx = torch.empty_like(x)
                    with _packs_of(ctx.pack_ident):
                        _lib.check(L.eemop_conv2d_bwd_data(dpre.data_ptr(), w.data_ptr(), n, hin, win, cin, c0, cs[i], cout, kh, kw, stride, ph, pw,
                                                           dx.data_ptr(), s))
                    dxs[i] = dx
                c0 += cs[i]
            if joined is not None:                                         # join before autograd hands dw / db on
                torch.cuda.current_stream(w.device).wait_event(joined)
        return (dw if (need[0] and ret_dw) else None, db if (has_b and need[1] and ret_db) else None, None, None, None, None, *dxs)


def conv2d(conv, *xs, act=ACT_NONE, out_scale=1.0, weight=None, bias=None):
    """Apply an nn.Conv2d parameter container (or explicit weight / bias slices of it) to channel-concatenated inputs."""
    w = conv.weight if weight is None else weight
    b = conv.bias if bias is None and weight is None else bias
    st = conv.stride[0]
    return Conv2d.apply(w, b, st, tuple(conv.padding), act, out_scale, *xs)


class InstanceNormReLU(torch.autograd.Function):
    """relu?(InstanceNorm2d(x)) - fnet's norm layers (model/extractor.py:31-35,43-57)."""

    @staticmethod
    def forward(ctx, x, relu):
        _need_cuda(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_instnorm_fwd(x.data_ptr(), None, n * c, h * w, 1 if relu else 0, y.data_ptr(), _sp(x)))
        ctx.save_for_backward(x, y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_instnorm_bwd(x.data_ptr(), y.data_ptr(), _c(dy).data_ptr(), n * c, h * w, 1 if ctx.relu else 0,
                                                     dx.data_ptr(), _sp(x)))
        return dx, None


class BatchNormTrainReLU(torch.autograd.Function):
    """relu?(BatchNorm2d(x)) with batch statistics; the module's running statistics are updated in place (momentum 0.1, unbiased
    variance) - cnet's norm layers with the module in train() (model/extractor.py:31-35; train_mvsec.py:231-235)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu):
        _need_cuda(x, weight)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(c, device=x.device)
        rstd = torch.empty(c, device=x.device)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_batchnorm_train_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(),
                                                            running_var.data_ptr(), n, c, h * w, float(momentum), float(eps), 1 if relu else 0,
                                                            y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _sp(x)))
        ctx.save_for_backward(x, y, weight, mean, rstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, rstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_batchnorm_train_bwd(x.data_ptr(), y.data_ptr(), _c(dy).data_ptr(), weight.data_ptr(), mean.data_ptr(),
                                                            rstd.data_ptr(), n, c, h * w, 1 if ctx.relu else 0, dx.data_ptr(), dw.data_ptr(),
                                                            db.data_ptr(), _sp(x)))
        return dx, dw, db, None, None, None, None, None


class BatchNormEvalReLU(torch.autograd.Function):
    """relu?(BatchNorm2d(x)) with the module in eval(): frozen running statistics, an affine map per channel whose weight and bias
    still train - cnet's norm layers after ERAFT.freeze_bn() (model/eraft.py:69-72)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, relu):
        _need_cuda(x, weight)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_batchnorm_eval_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(),
                                                           running_var.data_ptr(), n, c, h * w, float(eps), 1 if relu else 0, y.data_ptr(),
                                                           _sp(x)))
        ctx.save_for_backward(x, y, weight, running_mean, running_var)
        ctx.relu, ctx.eps = relu, float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, rm, rv = ctx.saved_tensors
        n, c, h, w = x.shape
        dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_batchnorm_eval_bwd(x.data_ptr(), y.data_ptr(), _c(dy).data_ptr(), weight.data_ptr(), rm.data_ptr(),
                                                           rv.data_ptr(), n, c, h * w, ctx.eps, 1 if ctx.relu else 0, dx.data_ptr(),
                                                           dw.data_ptr(), db.data_ptr(), _sp(x)))
        return dx, dw, db, None, None, None, None


def group_norm(x, num_groups, weight, bias, eps, relu):
    """relu?(GroupNorm(num_groups, C)(x)) (model/extractor.py:19-23,123-124) from two operators: the statistics of a group - its C / G
    channels are contiguous in NCHW - are InstanceNorm's over planes of (C / G) * H * W values, and the per-channel affine map (+ ReLU)
    is the frozen-BatchNorm operator with mean 0 and variance 1 - eps.  Both have their adjoints, so weight and bias train."""
    if abs(float(eps) - 1e-5) > 1e-12:
        raise ValueError("group_norm: eps other than 1e-5 is not built (the instance-norm operator's constant)")
    n, c, h, w = x.shape
    if c % num_groups:
        raise ValueError("group_norm: channels must divide into the groups")
    xn = InstanceNormReLU.apply(_c(x).view(n, num_groups, (c // num_groups) * h, w), False).view(n, c, h, w)
    zero = torch.zeros(c, device=x.device, dtype=torch.float32)
    var = torch.full((c,), 1.0 - float(eps), device=x.device, dtype=torch.float32)
    return BatchNormEvalReLU.apply(xn, weight, bias, zero, var, float(eps), relu)


def dropout2d(x, p):
    """nn.Dropout2d in training mode (model/extractor.py:147-149,183-184): whole channels zeroed with probability p, the rest scaled by
    1 / (1 - p).  The mask comes from torch's generator (as the reference's does); the multiply and its adjoint are the Mul operator."""
    n, c = x.shape[:2]
    keep = (torch.rand(n, c, 1, 1, device=x.device) >= p).to(torch.float32) / (1.0 - p)
    return Mul.apply(x, keep.expand_as(x).contiguous())


class AddReLU(torch.autograd.Function):
    """relu(x + y): the tail of a residual block (model/extractor.py:57)."""

    @staticmethod
    def forward(ctx, x, y):
        out = _binary(3, _c(x), _c(y))
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        d = _act_bwd(dout, out, ACT_RELU)
        return d, d


class Mul(torch.autograd.Function):
    """r * h (model/update.py:47,56)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        return _binary(2, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = _c(dout)
        return _binary(2, dout, b), _binary(2, dout, a)


class Add(torch.autograd.Function):
    """a + b (coords1 + delta_flow, model/eraft.py:149) / a - b with sign = -1 (coords1 - coords0, :144,155)."""

    @staticmethod
    def forward(ctx, a, b, sign):
        ctx.sign = sign
        return _binary(0 if sign > 0 else 1, _c(a), _c(b))

    @staticmethod
    def backward(ctx, dout):
        return dout, (dout if ctx.sign > 0 else _binary(4, _c(dout), alpha=-1.0)), None


def sum_tensors(ts):
    """Elementwise sum of a list of same-shaped CUDA tensors, eight per launch (eemop_sum_n)."""
    ts = [_c(t) for t in ts]
    while len(ts) > 1:
        head, ts = ts[:8], ts[8:]
        out = torch.empty_like(head[0])
        arr = (ctypes.c_void_p * len(head))(*[t.data_ptr() for t in head])
        with _on(out.device):
            _lib.check(_lib.lib().eemop_sum_n(ctypes.cast(arr, ctypes.c_void_p), len(head), out.numel(), out.data_ptr(), _sp(out)))
        ts.insert(0, out)
    return ts[0]


class FanOut(torch.autograd.Function):
    """n aliases of x, one per consumer.  A tensor with several consumers - E-RAFT's hidden state, context features, motion features and
    correlation pyramid across the twelve unrolled iterations (model/eraft.py:139-157) - otherwise gets its gradient from autograd's
    accumulation, n - 1 elementwise add launches (297 per E-RAFT training step); here the n gradients meet in one eemop_sum_n launch per
    eight of them.  The aliases must not be modified in place."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        return (sum_tensors(gs) if gs else None), None


def fan_out(x, n):
    """FanOut for tensors that take part in autograd; plain repetition otherwise (inference, constants)."""
    if n == 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return FanOut.apply(x, n)


class GRUBlend(torch.autograd.Function):
    """(1 - z) h + z q  (model/update.py:48,57)."""

    @staticmethod
    def forward(ctx, z, h, q):
        z, h, q = _c(z), _c(h), _c(q)
        out = torch.empty_like(h)
        with _on(h.device):
            _lib.check(_lib.lib().eemop_gru_blend(z.data_ptr(), h.data_ptr(), q.data_ptr(), h.numel(), out.data_ptr(), _sp(h)))
        ctx.save_for_backward(z, h, q)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, h, q = ctx.saved_tensors
        dz, dh, dq = torch.empty_like(z), torch.empty_like(h), torch.empty_like(q)
        with _on(h.device):
            _lib.check(_lib.lib().eemop_gru_blend_bwd(_c(dout).data_ptr(), z.data_ptr(), h.data_ptr(), q.data_ptr(), h.numel(), dz.data_ptr(),
                                                      dh.data_ptr(), dq.data_ptr(), _sp(h)))
        return dz, dh, dq


class Cat2(torch.autograd.Function):
    """torch.cat([a, b], dim=1) (model/update.py:79,81)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        n, ca, h, w = a.shape
        cb = b.shape[1]
        out = torch.empty(n, ca + cb, h, w, device=a.device, dtype=torch.float32)
        L = _lib.lib()
        with _on(a.device):
            _lib.check(L.eemop_copy_channels(a.data_ptr(), ca, 0, out.data_ptr(), ca + cb, 0, ca, n, h * w, _sp(a)))
            _lib.check(L.eemop_copy_channels(b.data_ptr(), cb, 0, out.data_ptr(), ca + cb, ca, cb, n, h * w, _sp(a)))
        ctx.split = (ca, cb)
        return out

    @staticmethod
    def backward(ctx, dout):
        ca, cb = ctx.split
        dout = _c(dout)
        n, _, h, w = dout.shape
        da = torch.empty(n, ca, h, w, device=dout.device, dtype=torch.float32)
        db = torch.empty(n, cb, h, w, device=dout.device, dtype=torch.float32)
        L = _lib.lib()
        with _on(dout.device):
            _lib.check(L.eemop_copy_channels(dout.data_ptr(), ca + cb, 0, da.data_ptr(), ca, 0, ca, n, h * w, _sp(dout)))
            _lib.check(L.eemop_copy_channels(dout.data_ptr(), ca + cb, ca, db.data_ptr(), cb, 0, cb, n, h * w, _sp(dout)))
        return da, db


class CorrPyramid(torch.autograd.Function):
    """CorrBlock.__init__ (model/corr.py:13-27,53-60): four pyramid levels of the all-pairs correlation."""

    @staticmethod
    def forward(ctx, f1, f2):
        f1, f2 = _c(f1), _c(f2)
        b, c, h, w = f1.shape
        lv = [torch.empty(b * h * w, 1, h >> l, w >> l, device=f1.device, dtype=torch.float32) for l in range(4)]
        with _on(f1.device):
            _lib.check(_lib.lib().eemop_corr_pyramid_fwd(f1.data_ptr(), f2.data_ptr(), b, c, h, w, *[t.data_ptr() for t in lv], _sp(f1)))
        ctx.save_for_backward(f1, f2)
        return tuple(lv)

    @staticmethod
    def backward(ctx, d0, d1, d2, d3):
        f1, f2 = ctx.saved_tensors
        b, c, h, w = f1.shape
        shapes = [(b * h * w, 1, h >> l, w >> l) for l in range(4)]
        ds = [(_c(d).clone() if d is not None else torch.zeros(s, device=f1.device)) for d, s in zip((d0, d1, d2, d3), shapes)]
        df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
        with _on(f1.device):
            _lib.check(_lib.lib().eraft_corr_pyramid_bwd(f1.data_ptr(), f2.data_ptr(), ds[0].data_ptr(), ds[1].data_ptr(), ds[2].data_ptr(),
                                                         ds[3].data_ptr(), b, c, h, w, df1.data_ptr(), df2.data_ptr(), _sp(f1)))
        return df1, df2


class CorrLookup(torch.autograd.Function):
    """CorrBlock.__call__ (model/corr.py:29-50); the coordinates carry no gradient (model/eraft.py:141 detaches them)."""

    @staticmethod
    def forward(ctx, coords, p0, p1, p2, p3):
        coords = _c(coords)
        b, _, h, w = coords.shape
        out = torch.empty(b, 324, h, w, device=coords.device, dtype=torch.float32)
        with _on(coords.device):
            _lib.check(_lib.lib().eemop_corr_lookup_fwd(p0.data_ptr(), p1.data_ptr(), p2.data_ptr(), p3.data_ptr(), coords.data_ptr(), b, h, w,
                                                        out.data_ptr(), _sp(coords)))
        ctx.save_for_backward(coords)
        ctx.shapes = [tuple(p.shape) for p in (p0, p1, p2, p3)]
        return out

    @staticmethod
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        b, _, h, w = coords.shape
        ds = [torch.empty(s, device=coords.device, dtype=torch.float32) for s in ctx.shapes]      # zeroed by the library
        with _on(coords.device):
            _lib.check(_lib.lib().eraft_corr_lookup_bwd(coords.data_ptr(), _c(dout).data_ptr(), b, h, w, *[d.data_ptr() for d in ds], _sp(coords)))
        return (None, *ds)


class ConvexUpsample(torch.autograd.Function):
    """ERAFT.upsample_flow (model/eraft.py:83-94) followed by InputPadder.unpad (utils/image_utils.py:142-145)."""

    @staticmethod
    def forward(ctx, flow, mask, pad):
        flow, mask = _c(flow), _c(mask)
        b, _, h, w = flow.shape
        zeros = torch.zeros_like(flow)
        full = torch.empty(b, 2, 8 * h, 8 * w, device=flow.device, dtype=torch.float32)
        with _on(flow.device):
            _lib.check(_lib.lib().eemop_convex_upsample_fwd(zeros.data_ptr(), flow.data_ptr(), mask.data_ptr(), b, h, w, full.data_ptr(), _sp(flow)))
        ctx.save_for_backward(flow, mask)
        ctx.pad = pad
        left, right, top, bottom = pad
        return full[:, :, top:8 * h - bottom, left:8 * w - right]

    @staticmethod
    def backward(ctx, dout):
        flow, mask = ctx.saved_tensors
        b, _, h, w = flow.shape
        left, right, top, bottom = ctx.pad
        dfull = torch.zeros(b, 2, 8 * h, 8 * w, device=flow.device, dtype=torch.float32)
        dfull[:, :, top:8 * h - bottom, left:8 * w - right] = dout          # placement of the cropped gradient: data movement only
        dflow, dmask = torch.empty_like(flow), torch.empty_like(mask)
        with _on(flow.device):
            _lib.check(_lib.lib().eraft_convex_upsample_bwd(flow.data_ptr(), mask.data_ptr(), dfull.data_ptr(), b, h, w, dflow.data_ptr(),
                                                            dmask.data_ptr(), _sp(flow)))
        return dflow, dmask, None


def coords_grids(batch, h, w, device, flow_init=None):
    """ERAFT.initialize_flow (model/eraft.py:73-81): coords0, coords1 [batch][2][h][w] (channel 0 = x, 1 = y), coords1 += flow_init."""
    c0 = torch.empty(batch, 2, h, w, device=device, dtype=torch.float32)
    c1 = torch.empty_like(c0)
    fi = _c(flow_init.float()) if flow_init is not None else None
    with _on(device):
        _lib.check(_lib.lib().eemop_coords_init(c0.data_ptr(), c1.data_ptr(), _ptr(fi), batch, h, w, _lib.current_stream_ptr(device)))
    return c0, c1


def sub(a, b):
    """a - b without a graph (both operands are constants of the graph: coords1 is detached, model/eraft.py:141-144)."""
    return _binary(1, _c(a), _c(b))


def replicate_pad_into(x, pad, out):
    """InputPadder.pad of x into the preallocated `out` (a batch slice of the encoder input)."""
    x = _c(x.float())
    n, c, h, w = x.shape
    left, right, top, bottom = pad
    assert tuple(out.shape) == (n, c, h + top + bottom, w + left + right) and out.is_contiguous()
    with _on(x.device):
        _lib.check(_lib.lib().eemop_replicate_pad(x.data_ptr(), out.data_ptr(), n * c, h, w, left, right, top, bottom, _sp(x)))
    return out


def replicate_pad(x, pad):
    """InputPadder.pad for one tensor (utils/image_utils.py:139-140); inputs carry no gradient."""
    x = _c(x.float())
    n, c, h, w = x.shape
    left, right, top, bottom = pad
    out = torch.empty(n, c, h + top + bottom, w + left + right, device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.check(_lib.lib().eemop_replicate_pad(x.data_ptr(), out.data_ptr(), n * c, h, w, left, right, top, bottom, _sp(x)))
    return out


# ------------------------------------------------------------------------------------------------ EEMFlow+ (EEMFlow_cdc) operators
ACT_LEAKY = 4


class ChannelSlice(torch.autograd.Function):
    """x[:, off:off + cnt] as a dense tensor (the input of one group of a grouped conv, x_out[:, :2] / [:, 2:3] of the upsampler)."""

    @staticmethod
    def forward(ctx, x, off, cnt):
        x = _c(x)
        n, c, h, w = x.shape
        out = torch.empty(n, cnt, h, w, device=x.device, dtype=torch.float32)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_copy_channels(x.data_ptr(), c, off, out.data_ptr(), cnt, 0, cnt, n, h * w, _sp(x)))
        ctx.cfg = (c, off, cnt)
        return out

    @staticmethod
    def backward(ctx, dout):
        c, off, cnt = ctx.cfg
        dout = _c(dout)
        n, _, h, w = dout.shape
        dx = torch.zeros(n, c, h, w, device=dout.device, dtype=torch.float32)
        with _on(dout.device):
            _lib.check(_lib.lib().eemop_copy_channels(dout.data_ptr(), cnt, 0, dx.data_ptr(), c, off, cnt, n, h * w, _sp(dout)))
        return dx, None, None


class CatN(torch.autograd.Function):
    """torch.cat(xs, dim=1)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(x) for x in xs]
        n, _, h, w = xs[0].shape
        cs = [x.shape[1] for x in xs]
        out = torch.empty(n, sum(cs), h, w, device=xs[0].device, dtype=torch.float32)
        L = _lib.lib()
        with _on(out.device):
            off = 0
            for x, c in zip(xs, cs):
                _lib.check(L.eemop_copy_channels(x.data_ptr(), c, 0, out.data_ptr(), sum(cs), off, c, n, h * w, _sp(out)))
                off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        n, ct, h, w = dout.shape
        L = _lib.lib()
        outs, off = [], 0
        with _on(dout.device):
            for i, c in enumerate(ctx.cs):
                if ctx.needs_input_grad[i]:
                    d = torch.empty(n, c, h, w, device=dout.device, dtype=torch.float32)
                    _lib.check(L.eemop_copy_channels(dout.data_ptr(), ct, off, d.data_ptr(), c, 0, c, n, h * w, _sp(dout)))
                    outs.append(d)
                else:
                    outs.append(None)
                off += c
        return tuple(outs)


class ChannelShuffle(torch.autograd.Function):
    """channel_shuffle(x, groups) (EEMFlow+.py:52-58)."""

    @staticmethod
    def forward(ctx, x, groups):
        x = _c(x)
        n, c, h, w = x.shape
        out = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_shuffle_channels(x.data_ptr(), out.data_ptr(), n, c, groups, h * w, 0, _sp(x)))
        ctx.groups = groups
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        n, c, h, w = dout.shape
        dx = torch.empty_like(dout)
        with _on(dout.device):
            _lib.check(_lib.lib().eemop_shuffle_channels(dout.data_ptr(), dx.data_ptr(), n, c, ctx.groups, h * w, 1, _sp(dout)))
        return dx, None


class AvgPool2(torch.autograd.Function):
    """F.avg_pool2d(x, 2, 2) (EEMFlow+.py:170-175)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n, c, h, w = x.shape
        out = torch.empty(n, c, h // 2, w // 2, device=x.device, dtype=torch.float32)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_pool2_fwd(x.data_ptr(), out.data_ptr(), n * c, h, w, _sp(x)))
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, c, h, w = ctx.shape
        dout = _c(dout)
        dx = torch.empty(n, c, h, w, device=dout.device, dtype=torch.float32)
        with _on(dout.device):
            _lib.check(_lib.lib().eemop_pool2_bwd(dout.data_ptr(), dx.data_ptr(), n * c, h, w, _sp(dout)))
        return dx


class LocalCorr53(torch.autograd.Function):
    """The 53 selected taps of the 9x9 local correlation, / C (EEMFlow+.py:14-23 + index_select)."""

    @staticmethod
    def forward(ctx, f1, f2):
        f1, f2 = _c(f1), _c(f2)
        b, c, h, w = f1.shape
        out = torch.empty(b, 53, h, w, device=f1.device, dtype=torch.float32)
        with _on(f1.device):
            _lib.check(_lib.lib().eemflow_local_corr53(f1.data_ptr(), f2.data_ptr(), b, c, h, w, out.data_ptr(), _sp(f1)))
        ctx.save_for_backward(f1, f2)
        return out

    @staticmethod
    def backward(ctx, dout):
        f1, f2 = ctx.saved_tensors
        b, c, h, w = f1.shape
        d1, d2 = torch.empty_like(f1), torch.empty_like(f2)
        with _on(f1.device):
            _lib.check(_lib.lib().eemop_local_corr53_bwd(_c(dout).data_ptr(), f1.data_ptr(), f2.data_ptr(), b, c, h, w, d1.data_ptr(),
                                                         d2.data_ptr(), _sp(f1)))
        return d1, d2


class Warp(torch.autograd.Function):
    """Backward bilinear warp; mode 0 EEMFlow_cdc.warp, 1 torch_warp, 2 WarpingLayer_no_div (the `>= 1` mask carries no gradient)."""

    @staticmethod
    def forward(ctx, x, flow, mode):
        x, flow = _c(x), _c(flow)
        b, c, h, w = x.shape
        out = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemplus_warp(x.data_ptr(), flow.data_ptr(), b, c, h, w, mode, out.data_ptr(), _sp(x)))
        ctx.save_for_backward(x, flow)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dout):
        x, flow = ctx.saved_tensors
        b, c, h, w = x.shape
        dx, dflow = torch.empty_like(x), torch.empty_like(flow)
        with _on(x.device):
            _lib.check(_lib.lib().eemplus_warp_bwd(x.data_ptr(), flow.data_ptr(), _c(dout).data_ptr(), b, c, h, w, ctx.mode, dx.data_ptr(),
                                                   dflow.data_ptr(), _sp(x)))
        return dx, dflow, None


class UpsampleFlowAs(torch.autograd.Function):
    """upsample2d_flow_as(inputs, target, 'bilinear', if_rate=True) (cdc_utils.py:80-103) as a pure function: returns (res, scaled) where
    `scaled` is what the reference leaves in `inputs` by its in-place multiplication (the caller continues with it)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        x = _c(x)
        n, _, h, w = x.shape
        su, sv = ow / w, oh / h
        L = _lib.lib()
        up = torch.empty(n, 2, oh, ow, device=x.device, dtype=torch.float32)
        res, scaled = torch.empty_like(up), torch.empty_like(x)
        with _on(x.device):
            s = _sp(x)
            _lib.check(L.eemop_resize_ac_fwd(x.data_ptr(), up.data_ptr(), n * 2, h, w, oh, ow, s))
            _lib.check(L.eemop_scale_flow(up.data_ptr(), n, oh * ow, su, sv, res.data_ptr(), s))
            _lib.check(L.eemop_scale_flow(x.data_ptr(), n, h * w, su, sv, scaled.data_ptr(), s))
        ctx.cfg = (n, h, w, oh, ow, su, sv)
        return res, scaled

    @staticmethod
    def backward(ctx, dres, dscaled):
        n, h, w, oh, ow, su, sv = ctx.cfg
        L = _lib.lib()
        dev = dres.device if dres is not None else dscaled.device
        dx = torch.zeros(n, 2, h, w, device=dev, dtype=torch.float32)
        with _on(dev):
            s = _lib.current_stream_ptr(dev)
            if dres is not None:
                t = torch.empty(n, 2, oh, ow, device=dev, dtype=torch.float32)
                _lib.check(L.eemop_scale_flow(_c(dres).data_ptr(), n, oh * ow, su, sv, t.data_ptr(), s))
                _lib.check(L.eemop_resize_ac_bwd(t.data_ptr(), dx.data_ptr(), n * 2, h, w, oh, ow, s))
            if dscaled is not None:
                t2 = torch.empty(n, 2, h, w, device=dev, dtype=torch.float32)
                _lib.check(L.eemop_scale_flow(_c(dscaled).data_ptr(), n, h * w, su, sv, t2.data_ptr(), s))
                dx = _binary(0, dx, t2)
        return dx, None, None


class Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        out = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.lib().eemop_act_fwd(x.data_ptr(), x.numel(), ACT_SIGMOID, out.data_ptr(), _sp(x)))
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        return _act_bwd(dout, out, ACT_SIGMOID)
