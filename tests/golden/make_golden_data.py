#!/usr/bin/env python3
"""Golden vectors for the dataset front-end rows and A15 - produced by EXECUTING THE
REFERENCE'S OWN FUNCTION SOURCES (build container only; the GPU box sees only tests/golden/data_rows.npz).

The reference modules cannot be imported here (cv2, imageio, matplotlib, torchvision, git are absent), so the
function definitions are taken out of the reference files with `ast` at run time and executed unmodified:
  loader/loader_utils.py : get_compressed_events, read_flo
  loader/HREM.py         : check_out_bounds, motion_propagate   (cv2.copyMakeBorder(..., BORDER_REPLICATE) is
                           provided as numpy.pad(mode="edge") - the one stand-in, for a package the image lacks)
  test_mvsec.py          : Test.flow_error
Nothing of the reference is copied into the repository; inputs are regenerated from the stored seeds.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_data.py
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
from eemflow_amd.hrem import (flow_error_inputs, synthetic_flow, synthetic_hrem_events, write_events_npz,  # noqa: E402
                              write_flo)


def ref_functions(path, names, env):
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, env)
            out[node.name] = env[node.name]
    assert set(out) == set(names), (names, list(out))
    return out


def main():
    cv2 = types.SimpleNamespace(BORDER_REPLICATE=1,
                                copyMakeBorder=lambda a, t, b, l, r, mode: np.pad(a, ((t, b), (l, r)), mode="edge"))
    env = {"numpy": np, "np": np, "cv2": cv2, "torch": torch, "print": print}
    lu = ref_functions(f"{REF}/loader/loader_utils.py", {"get_compressed_events", "read_flo"}, dict(env))
    hr = ref_functions(f"{REF}/loader/HREM.py", {"check_out_bounds", "motion_propagate"}, dict(env))
    fe = ref_functions(f"{REF}/test_mvsec.py", {"flow_error"}, dict(env))["flow_error"]

    out = {}
    with tempfile.TemporaryDirectory() as td:
        # ---- events npz + flo round trip through the reference readers
        ev = synthetic_hrem_events(3, 5000, 720, 1280)
        write_events_npz(os.path.join(td, "events1.npz"), ev)
        got = lu["get_compressed_events"](os.path.join(td, "events1.npz"))
        out["ev_seed"], out["ev_n"], out["ev_ref"] = 3, 5000, got
        fl = synthetic_flow(4, 96, 128)
        write_flo(os.path.join(td, "flow.flo"), fl)
        out["flo_seed"], out["flo_hw"], out["flo_ref"] = 4, np.array([96, 128]), lu["read_flo"](os.path.join(td, "flow.flo"))
    # ---- meshflow
    for tag, seed, h, w in (("a", 5, 720, 1280), ("b", 6, 260, 346), ("c", 7, 100, 150)):
        f = synthetic_flow(seed, h, w)
        xm, ym = hr["motion_propagate"](f.copy(), h, w)
        out[f"mp_{tag}_seed"], out[f"mp_{tag}_hw"] = seed, np.array([h, w])
        out[f"mp_{tag}_x"], out[f"mp_{tag}_y"] = xm, ym
    # ---- flow_error: dense / sparse, is_car, inf and zero-flow pixels, an all-equal case (EE sum == 0)
    h, w = 260, 346
    gt, pred, ev_img = flow_error_inputs(9, h, w)
    cases = []
    for et in ("dense", "sparse"):
        for car in (False, True):
            self = types.SimpleNamespace(data_loader=types.SimpleNamespace(dataset=types.SimpleNamespace(evaluation_type=et)))
            r = fe(self, torch.from_numpy(gt)[None], torch.from_numpy(pred)[None], torch.from_numpy(ev_img), is_car=car)
            cases.append([float(x) for x in r])
    self = types.SimpleNamespace(data_loader=types.SimpleNamespace(dataset=types.SimpleNamespace(evaluation_type="dense")))
    gz = np.nan_to_num(gt, posinf=1.0)
    r = fe(self, torch.from_numpy(gz)[None], torch.from_numpy(gz.copy())[None], torch.from_numpy(ev_img), is_car=False)
    cases.append([float(x) for x in r])
    out["fe_seed"], out["fe_hw"] = 9, np.array([h, w])      # inputs are rebuilt from the seeds by flow_error_inputs()
    out["fe_cases"] = np.array(cases, dtype=np.float64)       # dense, dense+car, sparse, sparse+car, identical
    np.savez_compressed(os.path.join(HERE, "data_rows.npz"), **out)
    print("wrote data_rows.npz", os.path.getsize(os.path.join(HERE, "data_rows.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
