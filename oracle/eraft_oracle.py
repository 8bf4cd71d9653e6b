"""CPU ORACLE for the E-RAFT part of the hot path.  TEST INFRASTRUCTURE ONLY (see eemflow_oracle.py).

Restates model/eraft.py, model/corr.py, model/update.py, model/extractor.py and model/model_utils.py of
the reference with torch-CPU fp32 functional ops.  Parity status: PINNED by tests/test_oracle_golden.py
against vectors produced by running those reference files (tests/golden/make_golden.py); these files
import in the build container without third-party stubs (only the utils_luo.tools stub).
"""
import math

import torch
import torch.nn.functional as F

from .eemflow_oracle import input_padder_pad, replicate_pad, unpad

# The reference's `.float()` casts (model/corr.py:27,50, model_utils.py:27, eraft.py:117) go through this name.  float32, as there; a
# test may set float64 and pass float64 parameters and inputs to run the same graph in double precision - the arbiter when two float32
# results differ through ReLU units that sit within round-off of 0.
FLOAT = torch.float32


# ----------------------------------------------------------------------------- A9 BasicEncoder
def _norm(x, sd, prefix, norm_fn, training=False):
    """norm_fn 'instance': InstanceNorm2d(affine=False); 'batch': BatchNorm2d running stats (eval)
    or batch stats (training) - model/extractor.py:19-41,123-133."""
    if norm_fn == "none":
        return x                                               # nn.Sequential() - model/extractor.py:36-40,131-132
    if norm_fn == "instance":
        return F.instance_norm(x, eps=1e-5)
    if norm_fn == "group":                                       # model/extractor.py:19-23,123-124: num_groups = planes // 8
        w = sd[prefix + "weight"]
        return F.group_norm(x, w.shape[0] // 8, w, sd[prefix + "bias"], eps=1e-5)
    if norm_fn == "batch":
        return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"],
                            sd[prefix + "bias"], training=training, momentum=0.1, eps=1e-5)
    raise ValueError(norm_fn)


def residual_block(sd, p, x, norm_fn, stride, training=False):
    """model/extractor.py:7-57."""
    y = F.relu(_norm(F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), sd,
                     p + "norm1.", norm_fn, training))
    y = F.relu(_norm(F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), sd, p + "norm2.", norm_fn, training))
    if stride != 1:
        x = _norm(F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride), sd,
                  p + "norm3.", norm_fn, training)
    return F.relu(x + y)


def basic_encoder(sd, prefix, x, norm_fn, training=False):
    """model/extractor.py:119-190 (eval, dropout 0).  x may be a list (batch-concatenated, :170-174)."""
    is_list = isinstance(x, (list, tuple))
    if is_list:
        b = x[0].shape[0]
        x = torch.cat(list(x), 0)
    x = F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], stride=2, padding=3)
    x = F.relu(_norm(x, sd, prefix + "norm1.", norm_fn, training))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = residual_block(sd, f"{prefix}{layer}.0.", x, norm_fn, stride, training)
        x = residual_block(sd, f"{prefix}{layer}.1.", x, norm_fn, 1, training)
    x = F.conv2d(x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    return torch.split(x, b, 0) if is_list else x


# ----------------------------------------------------------------------------- A10 / A11 correlation
def corr_pyramid(fmap1, fmap2, num_levels=4):
    """CorrBlock.__init__ + corr - model/corr.py:13-27,53-60: list of (B*H*W, 1, h_l, w_l)."""
    b, d, h, w = fmap1.shape
    corr = torch.matmul(fmap1.view(b, d, h * w).transpose(1, 2), fmap2.view(b, d, h * w))
    corr = corr.view(b, h, w, 1, h, w) / torch.sqrt(torch.tensor(d).to(FLOAT))
    corr = corr.reshape(b * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(num_levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def bilinear_sampler(img, coords):
    """model/model_utils.py:7-21: pixel coordinates, align_corners=True, zeros outside."""
    h, w = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (w - 1) - 1
    yg = 2 * yg / (h - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def corr_lookup(pyr, coords, radius=4):
    """CorrBlock.__call__ - model/corr.py:29-50.  coords (B,2,H,W) -> (B, L*(2r+1)^2, H, W).
    Quirk kept: delta = (dy[i], dx[j]) is ADDED TO (x, y), so channel i*9+j samples x+(i-r), y+(j-r)."""
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    out = []
    for i, corr in enumerate(pyr):
        dx = torch.linspace(-r, r, 2 * r + 1)
        dy = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1)
        centroid = coords.reshape(b * h1 * w1, 1, 1, 2) / 2 ** i
        sampled = bilinear_sampler(corr, centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2))
        out.append(sampled.view(b, h1, w1, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().to(FLOAT)


# ----------------------------------------------------------------------------- A12 update block
def _conv(sd, name, x, padding):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def motion_encoder(sd, p, flow, corr):
    """BasicMotionEncoder - model/update.py:63-81."""
    cor = F.relu(_conv(sd, p + "convc1", corr, 0))
    cor = F.relu(_conv(sd, p + "convc2", cor, 1))
    flo = F.relu(_conv(sd, p + "convf1", flow, 3))
    flo = F.relu(_conv(sd, p + "convf2", flo, 1))
    out = F.relu(_conv(sd, p + "conv", torch.cat([cor, flo], 1), 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, p, h, x):
    """SepConvGRU - model/update.py:33-60."""
    for suffix, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, p + "convz" + suffix, hx, pad))
        r = torch.sigmoid(_conv(sd, p + "convr" + suffix, hx, pad))
        q = torch.tanh(_conv(sd, p + "convq" + suffix, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, p, net, inp, corr, flow):
    """BasicUpdateBlock.forward - model/update.py:97-106 -> (net, mask, delta_flow)."""
    motion = motion_encoder(sd, p + "encoder.", flow, corr)
    net = sep_conv_gru(sd, p + "gru.", net, torch.cat([inp, motion], 1))
    delta = _conv(sd, p + "flow_head.conv2", F.relu(_conv(sd, p + "flow_head.conv1", net, 1)), 1)
    mask = 0.25 * _conv(sd, p + "mask.2", F.relu(_conv(sd, p + "mask.0", net, 1)), 0)
    return net, mask, delta


# ----------------------------------------------------------------------------- A13 ERAFT forward
def coords_grid(batch, ht, wd):
    """model/model_utils.py:24-27: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], 0).to(FLOAT)[None].repeat(batch, 1, 1, 1)


def convex_upsample(flow, mask):
    """ERAFT.upsample_flow - model/eraft.py:83-94."""
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


def eraft_forward(sd, events1, events2, iters=12, flow_init=None, image_size=None, keep=False, bn_training=False):
    """ERAFT.forward - model/eraft.py:97-159.  Returns (list of flow predictions, stages).  bn_training: cnet's BatchNorm layers
    use batch statistics and update sd's running_mean / running_var in place (the module in train(), train_mvsec.py:231-235)."""
    h, w = events1.shape[-2:]
    pad = input_padder_pad(*(image_size or (h, w)), mode="chairs", eval_pad_rate=32)   # eraft.py:65-67
    im1, im2 = replicate_pad(events1, pad).contiguous(), replicate_pad(events2, pad).contiguous()
    fmap1, fmap2 = basic_encoder(sd, "fnet.", [im1, im2], "instance")
    pyr = corr_pyramid(fmap1.to(FLOAT), fmap2.to(FLOAT))
    cnet = basic_encoder(sd, "cnet.", im1, "batch", bn_training)
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    n, _, hp, wp = im1.shape
    coords0 = coords_grid(n, hp // 8, wp // 8)
    coords1 = coords_grid(n, hp // 8, wp // 8)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    preds = []
    st = {"pad": pad, "fmap1": fmap1, "fmap2": fmap2, "net0": net, "inp": inp}
    for it in range(iters):
        coords1 = coords1.detach()                                      # eraft.py:141
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        net, mask, delta = update_block(sd, "update_block.", net, inp, corr, flow)
        coords1 = coords1 + delta
        preds.append(unpad(convex_upsample(coords1 - coords0, mask), pad))
        if keep and it == 0:
            st.update(corr0=corr, net1=net, mask1=mask, delta1=delta)
    if keep:
        st.update(pyr=pyr, flow_low=coords1 - coords0)
    return preds, st
