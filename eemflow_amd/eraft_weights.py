"""Seeded weights for the E-RAFT modules (no checkpoint ships with the reference).

Values come from numpy's PCG64 stream: convs Kaiming-normal fan_out/ReLU as model/extractor.py:151-158
(update-block convs too - PyTorch's default init is not reproduced, any fixed weights pin parity), small
random biases, BatchNorm affine and running statistics perturbed so those paths are exercised.  The
aliased keys `*.downsample.1.*` (same module object as `*.norm3.*`, model/extractor.py:49-50) get
identical values."""
from collections import OrderedDict

import numpy as np


def seeded_from_shapes(shapes, seed=0):
    """shapes: ordered {key: tuple}.  Returns ordered {key: np.ndarray}."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key, shape in shapes.items():
        if ".downsample.1." in key:
            sd[key] = sd[key.replace(".downsample.1.", ".norm3.")].copy()
        elif key.endswith("num_batches_tracked"):
            sd[key] = np.zeros((), dtype=np.int64)
        elif key.endswith("running_mean"):
            sd[key] = rng.normal(0, 0.2, shape).astype(np.float32)
        elif key.endswith("running_var"):
            sd[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            sd[key] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
            if "flow_head.conv2" in key:
                # an untrained update block emits ~20 px per iteration and the 12-step recurrence turns chaotic
                # (fp32 reordering noise grows 15x per step); a small flow head keeps the iteration contractive,
                # as a trained network's is, so that parity tolerances mean something
                sd[key] *= np.float32(0.02)
        elif key.endswith("weight"):                 # norm affine scale
            sd[key] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        else:                                        # biases
            sd[key] = rng.normal(0, 0.05, shape).astype(np.float32)
            if "flow_head.conv2" in key:
                sd[key] *= np.float32(0.1)
    return sd
