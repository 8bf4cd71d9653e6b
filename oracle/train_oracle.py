"""CPU ORACLE for the training-step math of the EEMFlow path.  TEST INFRASTRUCTURE ONLY.

Restates train_mvsec.py:201-227 (sequence_loss), :178-183 (AdamW + OneCycleLR), :241-258 (step: GradScaler,
clip_grad_norm_) with torch-CPU ops; gradients come from torch autograd through the functional oracle
forward (oracle/eemflow_oracle.py).  PINNED against tests/golden/train_step.npz, which was produced by
running the reference module + the reference's own sequence_loss source in the build container."""
import torch

from . import eemflow_oracle as O

MAX_FLOW = 400   # train_mvsec.py:41


def sequence_loss(flow_preds, flow_gt, valid, gamma=0.8, max_flow=MAX_FLOW):
    """train_mvsec.py:201-227."""
    n = len(flow_preds)
    loss = 0.0
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    valid = (valid >= 0.5) & (mag < max_flow)
    for i in range(n):
        w = gamma ** (n - i - 1)
        loss = loss + w * (valid[:, None] * (flow_preds[i] - flow_gt).abs()).mean()
    epe = torch.sum((flow_preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[valid.view(-1)]
    metrics = {"epe": epe.mean().item(), "1px": (epe < 1).float().mean().item(),
               "3px": (epe < 3).float().mean().item(), "5px": (epe < 5).float().mean().item()}
    return loss, metrics


def loss_and_grads(sd, events1, events2, flow_gt, valid, image_size=None, gamma=0.8, out_size=None, groups=5):
    """Forward (train-mode shapes), loss, and d loss / d parameter for every tensor of the state dict."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    flow, _ = O.eemflow_forward(params, events1, events2, image_size=image_size, out_size=out_size, groups=groups)
    loss, metrics = sequence_loss([flow], flow_gt, valid, gamma)
    loss.backward()
    return float(loss), metrics, {k: v.grad.detach() for k, v in params.items()}, flow.detach()


def make_optimizer(params, lr, wdecay, eps, num_steps):
    """train_mvsec.py:178-183."""
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wdecay, eps=eps)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, lr, num_steps + 100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy="linear")
    return opt, sched


def train_steps(sd, batches, lr=1e-4, wdecay=5e-5, eps=1e-8, num_steps=100, clip=1.0, gamma=0.8, image_size=None):
    """A few optimisation steps exactly as train_mvsec.py:241-258 orders them (fp32: the GradScaler's power-of-two
    scale/unscale is exact and is omitted).  Returns (losses, lrs, final params)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt, sched = make_optimizer(list(params.values()), lr, wdecay, eps, num_steps)
    losses, lrs = [], []
    for (e1, e2, gt, valid) in batches:
        opt.zero_grad()
        flow, _ = O.eemflow_forward(params, e1, e2, image_size=image_size)
        loss, _ = sequence_loss([flow], gt, valid, gamma)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), clip)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        losses.append(float(loss))
    return losses, lrs, {k: v.detach() for k, v in params.items()}
