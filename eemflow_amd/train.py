"""Training step of EEMFlow on MI355X with the reference trainer's semantics.

Mirrors `train.fetch_optimizer` / `train.sequence_loss` / the body of `train.train_iters`
(train_mvsec.py:178-183,201-227,241-258): AdamW(lr, wdecay, eps) + OneCycleLR(pct_start=0.05, linear,
num_steps + 100), gamma-weighted L1 sequence loss with the `valid >= 0.5 & |gt| < 400` mask,
clip_grad_norm_(clip).  All arithmetic runs in libeemflow_hip.so (eemflow_forward_backward /
eemflow_optimizer_step); the host only evaluates the learning-rate schedule and, in data-parallel jobs,
all-reduces the flat gradient buffer over RCCL (one collective per step - the reference's nn.DataParallel,
train_EEMFlow_HREM.py:116-118, re-broadcasts all weights and gathers all outputs every step instead).
mixed_precision=True in the reference config only enables a GradScaler around fp32 math; its power-of-two
scale/unscale is exact and is not reproduced.
"""
import ctypes
import os

import torch

from . import _lib, parallel


class OneCycleLinear:
    """torch.optim.lr_scheduler.OneCycleLR(max_lr, total_steps, pct_start=0.05, anneal_strategy='linear',
    cycle_momentum=False) evaluated on the host (defaults div_factor=25, final_div_factor=1e4)."""

    def __init__(self, max_lr, total_steps, pct_start=0.05, div_factor=25.0, final_div_factor=1e4):
        self.max_lr, self.total = float(max_lr), int(total_steps)
        self.initial = self.max_lr / div_factor
        self.min_lr = self.initial / final_div_factor
        self.end1 = float(pct_start * self.total) - 1.0
        self.end2 = float(self.total - 1)

    def lr(self, step):
        if step > self.end2:
            raise ValueError(f"OneCycle: step {step} beyond total_steps {self.total}")
        if step <= self.end1:
            pct = step / self.end1 if self.end1 > 0 else 1.0
            return (self.max_lr - self.initial) * pct + self.initial
        pct = (step - self.end1) / (self.end2 - self.end1)
        return (self.min_lr - self.max_lr) * pct + self.max_lr


class _LossTerm(torch.autograd.Function):
    """weight * mean(valid * |flow - gt|) and its gradient by `eemflow_sequence_loss` (one kernel, no host sync)."""

    @staticmethod
    def forward(ctx, flow, flow_gt, valid, weight, stats):
        flow = flow.contiguous()
        b, _, h, w = flow.shape
        dflow = torch.empty_like(flow)
        before = stats[0].clone()
        with torch.cuda.device(flow.device):
            _lib.check(_lib.lib().eemflow_sequence_loss(flow.data_ptr(), flow_gt.data_ptr(), valid.data_ptr(), b, h, w, float(weight),
                                                        dflow.data_ptr(), stats.data_ptr(), _lib.current_stream_ptr(flow.device)))
        ctx.save_for_backward(dflow)
        return ((stats[0] - before) * (float(weight) / flow.numel())).float()

    @staticmethod
    def backward(ctx, g):
        (dflow,) = ctx.saved_tensors
        return dflow * g, None, None, None, None


def sequence_loss(flow_preds, flow_gt, valid, gamma=0.8, metrics=True):
    """`train.sequence_loss` (train_mvsec.py:201-227) on the GPU: returns (loss, metrics) with the reference's metric keys; the loss
    is a differentiable CUDA scalar, so `scaler.scale(loss).backward()` continues into the model's HIP backward pass.
    max_flow is the reference's constant 400 (train_mvsec.py:41).  metrics=False: no host read of the statistics (the metrics come
    back as the 6-double device tensor of the last prediction instead of a dict) - what a step captured into a HIP graph needs."""
    if not flow_preds[0].is_cuda:
        raise _lib.EEMFlowHipError("sequence_loss: CUDA (ROCm) tensors required - there is no CPU path")
    gt, va = flow_gt.contiguous().float(), valid.contiguous().float()
    n = len(flow_preds)
    loss = 0.0
    for i, flow in enumerate(flow_preds):
        if tuple(flow.shape) != tuple(gt.shape) or tuple(va.shape) != (gt.shape[0],) + tuple(gt.shape[2:]):
            raise ValueError(f"sequence_loss: prediction {tuple(flow.shape)}, flow_gt {tuple(gt.shape)}, valid {tuple(va.shape)}")
        stats = torch.zeros(6, device=gt.device, dtype=torch.float64)
        loss = loss + _LossTerm.apply(flow.float(), gt, va, gamma ** (n - i - 1), stats)
    if not metrics:
        return loss, stats
    st = stats.tolist()                                          # statistics of the last prediction (train_mvsec.py:218-226)
    cnt = st[2] if st[2] > 0 else float("nan")
    return loss, {"epe": st[1] / cnt, "1px": st[3] / cnt, "3px": st[4] / cnt, "5px": st[5] / cnt}


class EEMFlowTrainer:
    """One optimisation step per `step()` call; owns the flat gradient buffer and the schedule."""

    def __init__(self, model, lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=1000000, clip=1.0, gamma=0.8):
        self.model = model
        self.wdecay, self.eps, self.clip, self.gamma = wdecay, epsilon, clip, gamma
        self.schedule = OneCycleLinear(lr, num_steps + 100)
        self.iteration = 0
        self.grad = None

    def step(self, events1, events2, flow_gt, valid):
        """Returns (loss, metrics dict, flow prediction).  Tensors are CUDA fp32: events (B,5,H,W), flow_gt
        (B,2,H,W) at the model's training output size, valid (B,H,W)."""
        m = self.model
        if not events1.is_cuda:
            raise _lib.EEMFlowHipError("EEMFlowTrainer.step: CUDA (ROCm) tensors required - there is no CPU path")
        dev = events1.device
        ctx = m._context(dev)
        e1, e2 = events1.contiguous().float(), events2.contiguous().float()
        gt, va = flow_gt.contiguous().float(), valid.contiguous().float()
        b, _, h, w = e1.shape
        oh, ow = (16, 16) if m.out_mesh_size else (h, w)
        if tuple(gt.shape) != (b, 2, oh, ow) or tuple(va.shape) != (b, oh, ow):
            raise ValueError(f"flow_gt must be {(b, 2, oh, ow)} and valid {(b, oh, ow)}, got {tuple(gt.shape)}, {tuple(va.shape)}")
        n = sum(p.numel() for p in m.parameters())
        if self.grad is None or self.grad.device != dev:
            self.grad = torch.empty(n, device=dev, dtype=torch.float32)
        flow = torch.empty(b, 2, oh, ow, device=dev, dtype=torch.float32)
        stats = (ctypes.c_double * 5)()
        L = _lib.lib()
        with torch.cuda.device(dev):
            s = _lib.current_stream_ptr(dev)
            sync_stats = os.environ.get("EEM_TRAIN_SYNC_STATS") == "1"       # A/B: the statistics read before the optimizer step is enqueued
            _lib.check(L.eemflow_forward_backward(ctx, e1.data_ptr(), e2.data_ptr(), gt.data_ptr(), va.data_ptr(), b, h, w, oh, ow,
                                                  1.0, flow.data_ptr(), self.grad.data_ptr(), ctypes.byref(stats) if sync_stats else None, s))
            # the loss statistics travel to the host behind an event; they are read after the optimizer step is enqueued (the GPU does
            # not wait for the host between the backward and the optimizer)
            if not sync_stats:
                _lib.check(L.eemflow_train_stats_async(ctx, s))
            parallel.average_gradients(self.grad)                # one RCCL all-reduce of 2.86 MB per step
            lr = self.schedule.lr(self.iteration)
            _lib.check(L.eemflow_optimizer_step(ctx, self.grad.data_ptr(), lr, self.wdecay, self.eps, self.clip, s))
            if not sync_stats:
                _lib.check(L.eemflow_train_stats_wait(ctx, stats))
        self.iteration += 1
        m._weights_on_device_are_newer = True
        return stats[0], {"epe": stats[1], "1px": stats[3], "3px": stats[4], "lr": lr}, flow

    def skipped_steps(self):
        """Optimizer steps skipped so far because the gradient held an inf or a NaN - GradScaler.step's behaviour in the reference loop
        (train_mvsec.py:237,257): weights, moments and bias corrections stay put, the learning-rate schedule advances."""
        m = self.model
        if getattr(m, "_ctx", None) is None:
            return 0
        out = ctypes.c_int(0)
        with torch.cuda.device(m._ctx_device):
            _lib.check(_lib.lib().eemflow_optimizer_skipped_steps(m._ctx, ctypes.byref(out), _lib.current_stream_ptr(m._ctx_device)))
        return out.value

    def sync_parameters(self):
        """Copy the device-resident weights back into the module's nn.Parameters (before state_dict()/checkpoint)."""
        m = self.model
        dev = m._ctx_device
        n = sum(p.numel() for p in m.parameters())
        flat = torch.empty(n, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().eemflow_get_weights(m._ctx, flat.data_ptr(), n, _lib.current_stream_ptr(dev)))
        off = 0
        with torch.no_grad():
            for v in m.state_dict().values():
                v.copy_(flat[off:off + v.numel()].view_as(v))
                off += v.numel()
        m._weights_version = m._weights_fingerprint()            # parameters now equal the device copy
        m._weights_on_device_are_newer = False
        return m
