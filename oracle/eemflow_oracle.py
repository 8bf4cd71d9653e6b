"""CPU ORACLE for the EEMFlow dense-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, op by op, what the reference computes on its PyTorch-CPU path.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
it - as the checker / the timed CPU baseline - never the product path
(eemflow_amd/ never imports oracle/).

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against vectors produced by running the reference's own files in the build container
(tests/golden/make_golden.py).  One caveat: the 9x9 local correlation is computed in the
reference by the third-party `spatial_correlation_sampler==0.4.0` (requirements.txt:131),
which is neither vendored nor installed; its goldens come from a restatement of its
documented semantics, corroborated in-tree by model/STEmodel/corr.py:14-26 and
model/IRRPWC/correlation_package/correlation_cuda_kernel.cu:82-108 (dy-major tap order).
That single op is therefore "parity unpinned" against the real extension.

All floating-point work is fp32 on torch CPU kernels (the reference's own CPU path is
torch ops); integer work (pad sizes, event-bin indices, tap lists) is numpy int64.
"""
import numpy as np
import torch
import torch.nn.functional as F

CORR_TAPS_53 = (
    0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 26, 28, 29, 30, 31, 32, 33, 34,
    36, 38, 39, 40, 41, 42, 44, 46, 47, 48, 49, 50, 51, 52, 54, 56, 57, 58, 59, 60, 62,
    64, 66, 68, 70, 72, 74, 76, 78, 80,
)   # model/EEMFlow/EEMFlow+.py:89-97 (see eemflow_amd/weights.py for why not EEMFlow.py:85-94)


# ----------------------------------------------------------------------------- A2 padding
def input_padder_pad(ht, wd, mode="chairs", eval_pad_rate=64):
    """[left, right, top, bottom] - utils/image_utils.py:129-137."""
    r = eval_pad_rate
    pad_ht = (((ht // r) + 1) * r - ht) % r
    pad_wd = (((wd // r) + 1) * r - wd) % r
    if mode == "sintel":
        return [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
    return [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]


def replicate_pad(x, pad):
    """utils/image_utils.py:139-140."""
    return F.pad(x, pad, mode="replicate")


def unpad(x, pad):
    """utils/image_utils.py:142-145."""
    ht, wd = x.shape[-2:]
    return x[..., pad[2]:ht - pad[3], pad[0]:wd - pad[1]]


def unpad_none(x):
    """EEMFlow / EEMFlow+ never crop: predictions are interpolated from the padded grid to the input size."""
    return x


# ----------------------------------------------------------------------------- A3 encoder
def convrelu(x, w, b, stride=1, groups=1):
    """3x3 conv, zero pad 1, bias, LeakyReLU(0.1) - model/EEMFlow/EEMFlow.py:26-30."""
    return F.leaky_relu(F.conv2d(x, w, b, stride=stride, padding=1, groups=groups), 0.1)


def encoder(sd, x):
    """EEMFlow.py:75-82,135-140: returns the three stage outputs (1/2, 1/4, 1/8)."""
    def cr(name, t, stride):
        return convrelu(t, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], stride)
    f1 = cr("pconv1_2", cr("pconv1_1", x, 2), 1)
    f2 = cr("pconv2_3", cr("pconv2_2", cr("pconv2_1", f1, 2), 1), 1)
    f3 = cr("pconv3_3", cr("pconv3_2", cr("pconv3_1", f2, 2), 1), 1)
    return f1, f2, f3


# ----------------------------------------------------------------------------- A4 pooling
def stage_pool(f, k):
    """EEMFlow.py:144-154: avg_pool2d(k, stride k), floor semantics."""
    return F.avg_pool2d(f, kernel_size=(k, k), stride=(k, k))


# ----------------------------------------------------------------------------- A5 correlation
def local_corr81(x, y, radius=4):
    """cv[b, (dy+r)*(2r+1)+(dx+r), h, w] = (1/C) sum_c x[b,c,h,w] * y[b,c,h+dy,w+dx], 0 outside.
    EEMFlow.py:14-23 (`.view(b,-1,h,w) / c`) over SpatialCorrelationSampler(1, 9, 1, 0, 1)."""
    b, c, h, w = x.shape
    d = 2 * radius + 1
    out = torch.zeros(b, d * d, h, w, dtype=x.dtype)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            ys0, ys1 = max(0, -dy), min(h, h - dy)
            xs0, xs1 = max(0, -dx), min(w, w - dx)
            if ys0 >= ys1 or xs0 >= xs1:
                continue
            prod = x[:, :, ys0:ys1, xs0:xs1] * y[:, :, ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
            out[:, (dy + radius) * d + (dx + radius), ys0:ys1, xs0:xs1] = prod.sum(1)
    return out / c


def local_corr53(x, y):
    """EEMFlow.py:160: index_select of the 53 diamond taps."""
    return local_corr81(x, y)[:, list(CORR_TAPS_53)]


# ----------------------------------------------------------------------------- A6 decoder
def channel_shuffle(x, groups):
    """EEMFlow.py:51-57: out channel j*groups+g <- in channel g*(C/groups)+j."""
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, c, h, w)


def decoder(sd, prefix, x, groups=5):
    """EEMFlow.py:59-69."""
    def cr(j, t, g=1):
        return convrelu(t, sd[f"{prefix}conv{j}.0.weight"], sd[f"{prefix}conv{j}.0.bias"], 1, g)
    out = cr(1, x)
    if groups == 1:
        out = cr(4, cr(3, cr(2, out)))
    else:
        for j in (2, 3, 4):
            out = channel_shuffle(cr(j, out, groups), groups)
    out = cr(6, cr(5, out))
    return F.conv2d(out, sd[f"{prefix}conv7.weight"], sd[f"{prefix}conv7.bias"], padding=1)


# ----------------------------------------------------------------------------- A7 + whole forward
def upsample_flow(flow, size):
    """EEMFlow.py:118-120: bilinear, align_corners=False, no magnitude scaling, no crop."""
    return F.interpolate(flow, size=tuple(size), mode="bilinear", align_corners=False)


def eemflow_forward(sd, events1, events2, image_size=None, groups=5, out_size=None, keep=False):
    """EEMFlow.forward (EEMFlow.py:122-183), inference semantics.

    sd: {key: torch fp32 tensor} in the checkpoint layout.  `image_size` is what the harness
    passed to change_imagesize (defaults to the input size).  Returns (flow, stages dict).
    """
    h, w = events1.shape[-2:]
    pad = input_padder_pad(*(image_size or (h, w)), mode="chairs", eval_pad_rate=64)
    out_size = out_size or (h, w)
    p1, p2 = replicate_pad(events1, pad), replicate_pad(events2, pad)
    st = {"pad": pad}
    fa = encoder(sd, p1)
    fb = encoder(sd, p2)
    flows = []
    for k, ps in ((1, 32), (2, 16), (3, 8)):
        pa, pb = stage_pool(fa[k - 1], ps), stage_pool(fb[k - 1], ps)
        cv = local_corr53(pa, pb)
        r = convrelu(pa, sd[f"rconv_{k}.0.weight"], sd[f"rconv_{k}.0.bias"])
        fl = decoder(sd, f"decoder_{k}.", torch.cat([cv, r], 1), groups)
        flows.append(fl)
        if keep:
            st.update({f"pool1_{k}": pa, f"pool2_{k}": pb, f"cv_{k}": cv, f"r_{k}": r, f"flow_{k}": fl})
    coarse = F.conv2d(torch.cat(flows, 1), sd["out_conv.weight"], sd["out_conv.bias"])
    flow = upsample_flow(coarse, out_size)
    if keep:
        st.update(f11=fa[0], f12=fa[1], f13=fa[2], f21=fb[0], f22=fb[1], f23=fb[2], coarse=coarse)
    return flow, st


def to_torch_sd(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------- A1 voxelizer
def event_sequence(features, timestamp_multiplier=None, convert_to_relative=False):
    """EventSequence.__init__ - loader/loader_utils.py:352-397 (works on a copy)."""
    f = np.array(features, dtype=np.float64, copy=True)
    if f.shape[0] > 1 and not np.all(f[:-1, 0] <= f[1:, 0]):
        f = f[np.argsort(f[:, 0])]                           # :389-392 (argsort default = quicksort)
    if timestamp_multiplier is not None:
        f[:, 0] *= timestamp_multiplier
    if convert_to_relative:
        f[:, 0] -= f[:, 0].min()
    return f


def voxel_indices(features, num_bins, height, width):
    """Integer part of EventSequenceToVoxelGrid_Pytorch.__call__ - loader_utils.py:476-523.

    Returns (idx_left, val_left, idx_right, val_right): int64 flat indices x + y*W + bin*W*H of the
    two index_add_ calls (only the valid events, in event order) and their fp32 weights."""
    ev = np.asarray(features, dtype=np.float64)
    t0, t1 = ev[0, 0], ev[-1, 0]
    delta = t1 - t0
    if delta == 0:
        delta = 1.0
    ts = (num_bins - 1) * (ev[:, 0] - t0) / delta            # f64, :488
    xs = ev[:, 1].astype(np.int64)                           # .long() truncation, :490-491
    ys = ev[:, 2].astype(np.int64)
    pols = ev[:, 3].astype(np.float32)
    pols[pols == 0] = -1                                     # :493
    tis = np.floor(ts)
    tl = tis.astype(np.int64)
    dts = (ts - tis).astype(np.float32)                      # dts.float(), :499-500
    vals_left = pols * (np.float32(1.0) - dts)
    vals_right = pols * dts
    vl = (tis < num_bins) & (tis >= 0)                       # :502-503
    vr = ((tis + 1) < num_bins) & (tis >= 0)                 # :517-518
    il = xs[vl] + ys[vl] * width + tl[vl] * width * height
    ir = xs[vr] + ys[vr] * width + (tl[vr] + 1) * width * height
    return il, vals_left[vl], ir, vals_right[vr]


def voxelize(features, num_bins, height, width, normalize=True):
    """EventSequenceToVoxelGrid_Pytorch.__call__ - loader_utils.py:447-537.  `features` is the
    (N,4) f64 array an EventSequence holds (sorted; scaled; relative)."""
    grid = np.zeros(num_bins * height * width, dtype=np.float32)
    il, vl, ir, vr = voxel_indices(features, num_bins, height, width)
    np.add.at(grid, il, vl)                                  # index_add_, sequential in event order
    np.add.at(grid, ir, vr)
    grid = torch.from_numpy(grid).view(num_bins, height, width)
    if normalize:                                            # :527-535
        mask = torch.nonzero(grid, as_tuple=True)
        if mask[0].numel() > 0:
            mean = grid[mask].mean()
            std = grid[mask].std()                           # unbiased; NaN for a single voxel
            if std > 0:
                grid[mask] = (grid[mask] - mean) / std
            else:
                grid[mask] = grid[mask] - mean
    return grid.numpy()


# ----------------------------------------------------------------------------- A15 metric
def flow_error_dense(flow_gt, flow_pred):
    """Test.flow_error, evaluation_type == 'dense', is_car False - test_mvsec.py:291-346.
    flow_*: (2,H,W) numpy.  Returns (AEE, %<1px, %<3px-or-10%, n_points)."""
    gt = np.transpose(flow_gt, (1, 2, 0))
    pr = np.transpose(flow_pred, (1, 2, 0))
    mask = (~np.isinf(gt[:, :, 0])) & (~np.isinf(gt[:, :, 1])) & (np.linalg.norm(gt, axis=2) > 0)
    g, p = gt[mask, :], pr[mask, :]
    ee = np.linalg.norm(g - p, axis=-1)
    ee_gt = np.linalg.norm(g, axis=-1)
    n = ee.shape[0]
    p1 = float((ee < 1.0).sum() / float(n + 1e-5))
    p3 = float(((ee < 3.0) | (ee < 0.1 * ee_gt)).sum()) / float(n + 1e-5)
    aee = 0.0 if ee.sum() == 0 else float(ee.mean())
    return aee, p1, p3, n
