"""Evaluation metrics on the GPU: the reference's Test.flow_error (test_mvsec.py:291-346) behind its call shape.

`flow_error(flow_gt, flow_pred, event_img, is_car=False, evaluation_type='dense')` takes the (1,2,H,W) CUDA tensors the
harness holds and returns the reference's 7-tuple (AEE, percent_1_AEE, percent_3_AEE, n_points, AEE_sum, AEE_gt,
AEE_gt_sum); the reduction runs in libeemflow_hip.so (eemflow_flow_error).  CUDA tensors only."""
import torch

from . import _lib


def flow_error_sums(flow_gt, flow_pred, event_img=None, is_car=False, evaluation_type="dense"):
    """The reduction only, no host synchronisation: a device tensor of 5 doubles (sum EE, sum |gt|, n, count(EE < 1),
    count(EE < 3 or EE < 0.1 |gt|)) on the current stream; `flow_error_from_sums` turns it into the reference's 7-tuple."""
    if not (flow_gt.is_cuda and flow_pred.is_cuda):
        raise _lib.EEMFlowHipError("flow_error: inputs must be CUDA (ROCm) tensors - there is no CPU path")
    gt = flow_gt[0].contiguous().float()
    pr = flow_pred[0].contiguous().float()
    _, h, w = gt.shape
    max_row = 190 if is_car else w                         # the reference crops rows with shape[1] = the WIDTH (:296)
    ev = None
    if evaluation_type == "sparse":
        ev = event_img.to(gt.device).reshape(h, w).contiguous().float()
    elif evaluation_type != "dense":
        raise ValueError(f"evaluation_type {evaluation_type!r}")
    out = torch.empty(5, dtype=torch.float64, device=gt.device)
    with torch.cuda.device(gt.device):
        _lib.check(_lib.lib().eemflow_flow_error(gt.data_ptr(), pr.data_ptr(), ev.data_ptr() if ev is not None else None, h, w,
                                                 max_row, out.data_ptr(), _lib.current_stream_ptr(gt.device)))
    return out


def flow_error_sums_many(flow_gts, flow_preds, event_imgs=None, is_car=False, evaluation_type="dense"):
    """`[flow_error_sums(g, p, e) for ...]` for up to 16 samples of one size by ONE launch (eemflow_flow_error_many): returns an (n, 5)
    float64 device tensor, row i = sample i's five sums."""
    import ctypes
    n = len(flow_gts)
    if not 1 <= n <= 16 or len(flow_preds) != n:
        raise ValueError("flow_error_sums_many: 1..16 samples, as many predictions as ground truths")
    if not all(t.is_cuda for t in list(flow_gts) + list(flow_preds)):
        raise _lib.EEMFlowHipError("flow_error: inputs must be CUDA (ROCm) tensors - there is no CPU path")
    gts = [g[0].contiguous().float() for g in flow_gts]
    prs = [p[0].contiguous().float() for p in flow_preds]
    _, h, w = gts[0].shape
    if any(tuple(t.shape) != (2, h, w) for t in gts + prs):
        raise ValueError("flow_error_sums_many: all samples share one (2,H,W) shape")
    max_row = 190 if is_car else w
    evs = None
    if evaluation_type == "sparse":
        evs = [e.to(gts[0].device).reshape(h, w).contiguous().float() for e in event_imgs]
    elif evaluation_type != "dense":
        raise ValueError(f"evaluation_type {evaluation_type!r}")
    dev = gts[0].device
    out = torch.empty(n, 5, dtype=torch.float64, device=dev)
    arr = ctypes.c_void_p * n
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().eemflow_flow_error_many(n, arr(*[t.data_ptr() for t in gts]), arr(*[t.data_ptr() for t in prs]),
                                                      arr(*[t.data_ptr() for t in evs]) if evs is not None else None, h, w, max_row,
                                                      out.data_ptr(), _lib.current_stream_ptr(dev)))
    return out


def flow_error_from_sums(sums):
    s_ee, s_gt, n, n1, n3 = sums.cpu().tolist() if torch.is_tensor(sums) else sums
    p1 = n1 / (n + 1e-5)
    p3 = n3 / (n + 1e-5)
    if s_ee == 0:
        return 0.0, p1, p3, int(n), 0.0, 0.0, 0.0
    return s_ee / n, p1, p3, int(n), s_ee, s_gt / n, s_gt


def flow_error(flow_gt, flow_pred, event_img=None, is_car=False, evaluation_type="dense"):
    return flow_error_from_sums(flow_error_sums(flow_gt, flow_pred, event_img, is_car, evaluation_type))
