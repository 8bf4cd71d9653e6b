"""Checkpoint layout of the EEMFlow dense-flow path and a seeded weight generator.

The layout (key names, shapes, order) is the reference's ``state_dict()`` for
``EEMFlow(config, groups=5, n_first_channels=5)``: 66 tensors (SURVEY.md says 58; the reference module registers 33 convs x {weight,bias}), 714 352 parameters
(reference: model/EEMFlow/EEMFlow.py:72-112; Decoder :37-46).  Keys may carry the
``module.`` prefix that ``nn.DataParallel`` adds (test_EEMFlow_HREM.py:62-66).

No checkpoint ships with the reference, so tests and the bench use weights drawn
here from numpy's PCG64 stream (stable across platforms and torch versions):
Kaiming-normal weights as in EEMFlow.py:108-112 and, for tests, small non-zero
biases so that the bias path is pinned too.
"""
from collections import OrderedDict

import numpy as np

# 53 correlation taps out of the 9x9 window, dy-major (reference:
# model/EEMFlow/EEMFlow+.py:89-97).  The list shipped in EEMFlow.py:85-94 has 49
# entries and cannot feed Decoder(69) (49+16 != 69); 53+16 == 69 is the only
# list consistent with the checkpoint layout.  See DESIGN.md "53-tap list".
CORR_TAPS_53 = (
    0, 2, 4, 6, 8,
    10, 12, 14, 16,
    18, 20, 21, 22, 23, 24, 26,
    28, 29, 30, 31, 32, 33, 34,
    36, 38, 39, 40, 41, 42, 44,
    46, 47, 48, 49, 50, 51, 52,
    54, 56, 57, 58, 59, 60, 62,
    64, 66, 68, 70,
    72, 74, 76, 78, 80,
)

ENCODER_LAYERS = (
    # name, cin, cout, stride       (EEMFlow.py:75-82)
    ("pconv1_1", None, 16, 2),
    ("pconv1_2", 16, 16, 1),
    ("pconv2_1", 16, 32, 2),
    ("pconv2_2", 32, 32, 1),
    ("pconv2_3", 32, 32, 1),
    ("pconv3_1", 32, 64, 2),
    ("pconv3_2", 64, 64, 1),
    ("pconv3_3", 64, 64, 1),
)


def eemflow_param_shapes(n_first_channels=5, groups=5):
    """Ordered {key: shape} exactly as the reference module registers them."""
    dec_in = len(CORR_TAPS_53) + 16          # 69
    dec_w = 100                              # Decoder width (EEMFlow.py:42-45)
    shapes = OrderedDict()
    for name, cin, cout, _ in ENCODER_LAYERS:
        cin = n_first_channels if cin is None else cin
        shapes[f"{name}.0.weight"] = (cout, cin, 3, 3)
        shapes[f"{name}.0.bias"] = (cout,)
    for k, cin in ((1, 16), (2, 32), (3, 64)):           # EEMFlow.py:96-98
        shapes[f"rconv_{k}.0.weight"] = (16, cin, 3, 3)
        shapes[f"rconv_{k}.0.bias"] = (16,)
    for k in (1, 2, 3):                                   # EEMFlow.py:100-102
        p = f"decoder_{k}."
        shapes[p + "conv1.0.weight"] = (dec_w, dec_in, 3, 3)
        shapes[p + "conv1.0.bias"] = (dec_w,)
        for j in (2, 3, 4):
            shapes[p + f"conv{j}.0.weight"] = (dec_w, dec_w // groups, 3, 3)
            shapes[p + f"conv{j}.0.bias"] = (dec_w,)
        shapes[p + "conv5.0.weight"] = (64, dec_w, 3, 3)
        shapes[p + "conv5.0.bias"] = (64,)
        shapes[p + "conv6.0.weight"] = (32, 64, 3, 3)
        shapes[p + "conv6.0.bias"] = (32,)
        shapes[p + "conv7.weight"] = (2, 32, 3, 3)
        shapes[p + "conv7.bias"] = (2,)
    shapes["out_conv.weight"] = (2, 6, 1, 1)              # EEMFlow.py:104
    shapes["out_conv.bias"] = (2,)
    return shapes


def seeded_state_dict(seed=0, n_first_channels=5, groups=5, bias_std=0.05):
    """numpy state dict: Kaiming-normal (fan_in, gain sqrt 2) weights, N(0, bias_std) biases.

    ``bias_std=0`` reproduces the reference init (zero biases, EEMFlow.py:111-112).
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key, shape in eemflow_param_shapes(n_first_channels, groups).items():
        if key.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        else:
            sd[key] = (rng.standard_normal(shape) * bias_std).astype(np.float32)
    return sd


def strip_module_prefix(state_dict):
    """Accept DataParallel checkpoints (reference: test_EEMFlow_HREM.py:64-66)."""
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def synthetic_voxel_pair(seed, batch, height, width, bins=5, density=0.2):
    """Synthetic event-voxel pair with the statistics the voxelizer's normalisation
    produces: ~`density` non-zero voxels, zero-mean unit-variance."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        v = rng.standard_normal((batch, bins, height, width), dtype=np.float32)
        m = rng.random((batch, bins, height, width), dtype=np.float32) < density
        out.append((v * m).astype(np.float32))
    return out


def synthetic_gt(seed, batch, h, w):
    """Smooth ground-truth flow + validity mask with holes and a few |gt| > 400 pixels (numpy PCG64);
    the same generator tests/golden/make_golden.py used for train_step.npz."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    gt = np.stack([3 * np.sin(2 * np.pi * xx / w + 0.3), 2 * np.cos(2 * np.pi * yy / h)])[None].repeat(batch, 0)
    gt = (gt + rng.normal(0, 0.5, gt.shape)).astype(np.float32)
    gt[:, :, 0, :3] = 500.0                                  # excluded by mag < MAX_FLOW
    valid = (rng.random((batch, h, w)) < 0.8).astype(np.float32)
    return gt, valid
