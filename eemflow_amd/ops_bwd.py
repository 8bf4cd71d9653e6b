"""Backward passes of the non-convolution operators behind torch.Tensor arguments.

  corr_lookup_bwd(coords, dout)                     autograd of CorrBlock.__call__ w.r.t. the pyramid (model/corr.py:29-50)
  corr_pyramid_bwd(fmap1, fmap2, dpyr)              autograd of CorrBlock.__init__ / corr            (model/corr.py:13-27,53-60)
  convex_upsample_bwd(flow, mask, dout)             autograd of ERAFT.upsample_flow                  (model/eraft.py:83-94)
  warp_bwd(x, flow, dout, mode)                     autograd of the three warps (EEMFlow+.py:137-149, cdc_utils.py:50-78,
                                                    utils_luo/tools.py:2262-2306)
All arithmetic runs in libeemflow_hip.so (csrc/bwd_ops.hip); CUDA tensors only."""
import torch

from . import _lib


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.EEMFlowHipError("backward ops need CUDA (ROCm) tensors - there is no CPU path")


def _c(t):
    return t.contiguous().float()


def corr_lookup_bwd(coords, dout):
    """coords (B,2,H,W), dout (B,324,H,W) -> [dpyr_l (B*H*W, 1, H>>l, W>>l) for l in 0..3]."""
    _need_cuda(coords, dout)
    coords, dout = _c(coords), _c(dout)
    b, _, h, w = coords.shape
    d = [torch.empty(b * h * w, 1, h >> l, w >> l, device=coords.device) for l in range(4)]
    with torch.cuda.device(coords.device):
        _lib.check(_lib.lib().eraft_corr_lookup_bwd(coords.data_ptr(), dout.data_ptr(), b, h, w, d[0].data_ptr(), d[1].data_ptr(),
                                                    d[2].data_ptr(), d[3].data_ptr(), _lib.current_stream_ptr(coords.device)))
    return d


def corr_pyramid_bwd(fmap1, fmap2, dpyr):
    """fmaps (B,C,H,W), dpyr as returned by corr_lookup_bwd (levels 0..2 are modified in place) -> (dfmap1, dfmap2)."""
    _need_cuda(fmap1, fmap2, *dpyr)
    f1, f2 = _c(fmap1), _c(fmap2)
    b, c, h, w = f1.shape
    d1, d2 = torch.empty_like(f1), torch.empty_like(f2)
    with torch.cuda.device(f1.device):
        _lib.check(_lib.lib().eraft_corr_pyramid_bwd(f1.data_ptr(), f2.data_ptr(), dpyr[0].data_ptr(), dpyr[1].data_ptr(),
                                                     dpyr[2].data_ptr(), dpyr[3].data_ptr(), b, c, h, w, d1.data_ptr(), d2.data_ptr(),
                                                     _lib.current_stream_ptr(f1.device)))
    return d1, d2


def convex_upsample_bwd(flow, mask, dout):
    """flow (B,2,H,W), mask (B,576,H,W), dout (B,2,8H,8W) -> (dflow, dmask)."""
    _need_cuda(flow, mask, dout)
    flow, mask, dout = _c(flow), _c(mask), _c(dout)
    b, _, h, w = flow.shape
    dflow, dmask = torch.empty_like(flow), torch.empty_like(mask)
    with torch.cuda.device(flow.device):
        _lib.check(_lib.lib().eraft_convex_upsample_bwd(flow.data_ptr(), mask.data_ptr(), dout.data_ptr(), b, h, w, dflow.data_ptr(),
                                                        dmask.data_ptr(), _lib.current_stream_ptr(flow.device)))
    return dflow, dmask


def warp_bwd(x, flow, dout, mode):
    """x, dout (B,C,H,W), flow (B,2,H,W); mode 0 EEMFlow_cdc.warp, 1 torch_warp, 2 WarpingLayer_no_div -> (dx, dflow)."""
    _need_cuda(x, flow, dout)
    x, flow, dout = _c(x), _c(flow), _c(dout)
    b, c, h, w = x.shape
    dx, dflow = torch.empty_like(x), torch.empty_like(flow)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().eemplus_warp_bwd(x.data_ptr(), flow.data_ptr(), dout.data_ptr(), b, c, h, w, mode, dx.data_ptr(),
                                               dflow.data_ptr(), _lib.current_stream_ptr(x.device)))
    return dx, dflow
