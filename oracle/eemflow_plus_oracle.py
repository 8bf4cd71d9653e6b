"""CPU ORACLE for EEMFlow+ (EEMFlow_cdc: the coarse-to-fine bilinear flow-warp loop).  TEST INFRASTRUCTURE ONLY.

Restates model/EEMFlow/EEMFlow+.py:74-234, model/EEMFlow/cdc_utils.py:50-177 and utils_luo/tools.py:2262-2306
(torch_warp) with torch-CPU fp32 ops.  PINNED against tests/golden/eemflow_plus_*.npz (reference run in the
build container; the unimportable utils_luo.tools is stubbed with a restated torch_warp, the un-vendored
spatial_correlation_sampler as in eemflow_oracle.py - the same "parity unpinned" caveat applies to that op).

Quirks kept on purpose (each changes numbers):
  * WarpingLayer_no_div and torch_warp normalise by (W-1),(H-1) but call grid_sample with its default
    align_corners=False (cdc_utils.py:68-71, tools.py:2285-2291): samples land half a pixel off; the mask is
    `grid_sample(ones) >= 1.0`, which also drops interior pixels whose four weights sum to 0.99999994.
  * EEMFlow_cdc.warp uses align_corners=True (EEMFlow+.py:148).
  * upsample2d_flow_as(if_rate=True) scales its INPUT in place after interpolating (cdc_utils.py:85-86), so
    flow6..flow3 are doubled by the next level's cdc_model call before the final full-resolution upsampling.
"""
import torch
import torch.nn.functional as F

from . import eemflow_oracle as O

TAPS = list(O.CORR_TAPS_53)


def conv_lrelu(sd, name, x, stride=1, groups=1, k=3, act=True):
    y = F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=(k - 1) // 2, groups=groups)
    return F.leaky_relu(y, 0.1) if act else y


def _grid(x, flo):
    b, _, h, w = x.shape
    xx = torch.arange(0, w).view(1, -1).repeat(h, 1).view(1, 1, h, w).repeat(b, 1, 1, 1)
    yy = torch.arange(0, h).view(-1, 1).repeat(1, w).view(1, 1, h, w).repeat(b, 1, 1, 1)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    vgrid[:, 0] = 2.0 * vgrid[:, 0] / max(w - 1, 1) - 1.0
    vgrid[:, 1] = 2.0 * vgrid[:, 1] / max(h - 1, 1) - 1.0
    return vgrid.permute(0, 2, 3, 1)


def warp_align_true(x, flo):
    """EEMFlow_cdc.warp - EEMFlow+.py:137-149."""
    return F.grid_sample(x, _grid(x, flo), mode="bilinear", align_corners=True)


def torch_warp(x, flo):
    """tensor_tools.torch_warp - utils_luo/tools.py:2262-2306."""
    return F.grid_sample(x, _grid(x, flo), padding_mode="zeros")


def warping_layer_no_div(x, flow):
    """WarpingLayer_no_div - cdc_utils.py:50-78."""
    g = _grid(x, flow)
    xw = F.grid_sample(x, g, padding_mode="zeros")
    mask = (F.grid_sample(torch.ones_like(x), g) >= 1.0).float()
    return xw * mask


def upsample2d_flow_as(inputs, target_hw, if_rate=False):
    """cdc_utils.py:80-103.  Mutates `inputs` in place when if_rate (kept!)."""
    h, w = target_hw
    res = F.interpolate(inputs, [h, w], mode="bilinear", align_corners=True)
    if if_rate:
        _, _, h_, w_ = inputs.shape
        inputs[:, 0] *= (w / w_)
        inputs[:, 1] *= (h / h_)
        u, v = res.chunk(2, dim=1)
        res = torch.cat([u * (w / w_), v * (h / h_)], dim=1)
    return res


def dense_estimator(sd, p, x):
    """FlowEstimatorDense_temp - cdc_utils.py:109-145."""
    for i in range(1, 6):
        x = torch.cat([conv_lrelu(sd, f"{p}conv{i}.0", x), x], dim=1)
    return x, conv_lrelu(sd, f"{p}conv_last.0", x, act=False)


def cdc_forward(sd, flow_init, f1, f2, keep=None):
    """cdc_model.forward with output_level_flow=None - cdc_utils.py:156-174 -> flow_up."""
    if flow_init.shape[-2:] != f1.shape[-2:]:
        flow_init = upsample2d_flow_as(flow_init, f1.shape[-2:], if_rate=True)
    if keep is not None:
        keep.append(flow_init.clone())
    f2w = warping_layer_no_div(f2, flow_init)
    _, x_out = dense_estimator(sd, "cdc_model.dense_estimator_mask.", torch.cat((f1, f2w), dim=1))
    inter_flow = x_out[:, :2]
    inter_mask = torch.sigmoid(x_out[:, 2:3])
    return torch_warp(flow_init, inter_flow) * (1 - inter_mask) + flow_init * inter_mask


def decoder(sd, p, x, groups=3):
    """Decoder - EEMFlow+.py:38-71 (width 96)."""
    out = conv_lrelu(sd, p + "conv1.0", x)
    for j in (2, 3, 4):
        out = conv_lrelu(sd, p + f"conv{j}.0", out, groups=groups)
        if groups != 1:
            out = O.channel_shuffle(out, groups)
    out = conv_lrelu(sd, p + "conv6.0", conv_lrelu(sd, p + "conv5.0", out))
    return F.conv2d(out, sd[p + "conv7.weight"], sd[p + "conv7.bias"], padding=1)


def corr53(x, y):
    """Same arithmetic as the golden generator's restatement of SpatialCorrelationSampler(1, 9, 1, 0, 1) (zero-pad
    y by 4, 81 shifted products summed over channels), so that the oracle is bit-identical with the generator:
    the warp masks downstream flip on 1-ulp differences."""
    b, c, h, w = x.shape
    yp = F.pad(y, (4, 4, 4, 4))
    out = [(x * yp[:, :, i:i + h, j:j + w]).sum(1) for i in range(9) for j in range(9)]
    return (torch.stack(out, 1).view(b, 9, 9, h, w).view(b, -1, h, w) / c)[:, TAPS]


def level_from_init(sd, l, f1l, f2l, flow_init, groups=3):
    """One l-block of EEMFlow_cdc.forward (EEMFlow+.py:183-229) from cdc_model's (already upsampled) flow_init:
    returns (flow_up_l, flow_l).  The teacher-forced unit the per-level parity tests compare."""
    a = conv_lrelu(sd, f"conv_1x1.{l}.0", f1l, k=1)
    b = conv_lrelu(sd, f"conv_1x1.{l}.0", f2l, k=1)
    flow_up = cdc_forward(sd, flow_init, a, b)
    f2w = warp_align_true(f2l, flow_up)
    cat = torch.cat([corr53(f1l, f2w), conv_lrelu(sd, f"rconv{l}.0", f1l), flow_up], 1)
    return flow_up, decoder(sd, f"decoder{l}.", cat, groups) + flow_up


def eemflow_plus_forward(sd, events1, events2, image_size=None, groups=3, keep=False):
    """EEMFlow_cdc.forward - EEMFlow+.py:158-234.  Returns ([5 full-resolution flows coarse->fine], stages)."""
    h, w = events1.shape[-2:]
    pad = O.input_padder_pad(*(image_size or (h, w)), mode="chairs", eval_pad_rate=64)
    i1, i2 = O.replicate_pad(events1, pad), O.replicate_pad(events2, pad)
    fa, fb = list(O.encoder(sd, i1)), list(O.encoder(sd, i2))           # f11,f12,f13 / f21,f22,f23
    for _ in range(3):
        fa.append(F.avg_pool2d(fa[-1], 2, 2))
        fb.append(F.avg_pool2d(fb[-1], 2, 2))
    f1 = {l: fa[l - 1] for l in range(1, 7)}
    f2 = {l: fb[l - 1] for l in range(1, 7)}
    flow_up = torch.zeros(f1[6].size(0), 2, f1[6].size(2), f1[6].size(3))
    cat = torch.cat([corr53(f1[6], f2[6]), conv_lrelu(sd, "rconv6.0", f1[6]), flow_up], 1)
    flows = {6: decoder(sd, "decoder6.", cat, groups)}
    st = {"pad": pad}
    for l in (5, 4, 3, 2):
        a = conv_lrelu(sd, f"conv_1x1.{l}.0", f1[l], k=1)
        b = conv_lrelu(sd, f"conv_1x1.{l}.0", f2[l], k=1)
        inits = [] if keep else None
        flow_up = cdc_forward(sd, flows[l + 1], a, b, inits)             # doubles flows[l+1] in place (quirk)
        f2w = warp_align_true(f2[l], flow_up)
        cat = torch.cat([corr53(f1[l], f2w), conv_lrelu(sd, f"rconv{l}.0", f1[l]), flow_up], 1)
        flows[l] = decoder(sd, f"decoder{l}.", cat, groups) + flow_up
        if keep:
            st[f"flow_up{l}"] = flow_up
            st[f"flow_init{l}"] = inits[0]
            st[f"flow_raw{l}"] = flows[l].clone()                        # before the next level doubles it in place
    if keep:
        st.update({f"flow{l}": flows[l].clone() for l in flows})
        st["f1"], st["f2"] = f1, f2
    preds = [O.unpad_none(upsample2d_flow_as(flows[l], (h, w), if_rate=True)) for l in (6, 5, 4, 3, 2)]
    return preds, st
