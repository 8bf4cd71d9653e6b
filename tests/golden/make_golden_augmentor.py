#!/usr/bin/env python3
"""Golden vectors for eemflow_amd/augmentor.py - produced by EXECUTING THE REFERENCE'S OWN CLASS SOURCES (build container
only).  utils/augumentor.py cannot be imported here (cv2, torchvision, PIL are absent), so the two class definitions are taken
out of the file with `ast` and executed unmodified, with `ColorJitter` (constructed, never called) provided as a no-op stand-in.
Nothing of the reference is copied into the repository; inputs are regenerated from the stored seeds.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_augmentor.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils/augumentor.py"


def ref_classes(names):
    tree = ast.parse(open(REF).read())
    env = {"np": np, "ColorJitter": lambda **kw: None}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), env)
    return [env[n] for n in names]


def inputs(seed, h, w, c):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((h, w, c)).astype(np.float32) for _ in range(4)] + [rng.standard_normal((h, w, 2))]


def main():
    Flow, Dense = ref_classes(["FlowAugmentor", "DenseSparseAugmentor"])
    out = {}
    cases = []
    for k, (seed, h, w, crop, flip) in enumerate([(1, 20, 28, (16, 24), True), (2, 20, 28, (20, 28), True), (3, 18, 22, (15, 20), False),
                                                  (4, 26, 34, (24, 24), True), (5, 17, 23, (8, 8), True), (6, 24, 32, (12, 16), True)]):
        a, b, da, db, fl = inputs(100 + seed, h, w, 3)
        np.random.seed(seed)
        f1 = Flow(crop_size=list(crop), do_flip=flip)(a, b, fl, without_resize=True)
        np.random.seed(seed)
        d = Dense(crop_size=list(crop), do_flip=flip)(a, b, da, db, fl)
        cases.append([seed, h, w, crop[0], crop[1], int(flip)])
        for i, arr in enumerate(f1):
            out[f"flow_nr_{k}_{i}"] = arr
        for i, arr in enumerate(d):
            out[f"dense_{k}_{i}"] = arr
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "augmentor.npz"), **out)
    print("wrote augmentor.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
