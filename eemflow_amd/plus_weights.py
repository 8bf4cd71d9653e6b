"""Seeded weights for EEMFlow+ (EEMFlow_cdc): Kaiming-normal convs, small biases, numpy PCG64 (no checkpoint
ships with the reference).  conv7 of every decoder and the dense estimator's last conv are scaled down so that
an untrained network produces sub-pixel flows: the reference's `grid_sample(ones) >= 1.0` warp mask flips on
ulp-level differences, and small flows keep that from dominating parity tests."""
from collections import OrderedDict

import numpy as np


def seeded_from_shapes(shapes, seed=0):
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key, shape in shapes.items():
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            w = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
            if "conv7" in key or "conv_last" in key:
                w *= np.float32(0.25)
            sd[key] = w
        else:
            sd[key] = rng.normal(0, 0.05, shape).astype(np.float32)
    return sd
