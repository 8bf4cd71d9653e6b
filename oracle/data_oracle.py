"""CPU restatement (TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product) of the rows of
SURVEY.md section 8(f) that sit either side of the hot path, and of A15:

  get_compressed_events   loader/loader_utils.py:26-42    HREM events{1,2}.npz -> (N,4) float64 [t s, x, y, p in {-1,+1}]
  read_flo                loader/loader_utils.py:54-65    Middlebury .flo -> (H,W,2) float32
  motion_propagate        loader/HREM.py:30-99            dense flow -> 16x16 mesh flow (vertex medians + 5x5 median)
  flow_error              test_mvsec.py:291-346           AEE / %1px / %3px statistics, dense | sparse, is_car crop

Pinned by tests/test_data_rows.py against tests/golden/data_rows.npz, produced by executing the reference's own
function sources (tests/golden/make_golden_data.py).  Loops follow the reference line by line on purpose.
"""
import numpy as np
from scipy.signal import medfilt2d


def get_compressed_events(event_path):
    d = np.load(event_path)                                               # loader_utils.py:28-32
    p = 2 * d["p"] - 1                                                    # :34
    return np.stack([d["t"] * 1e-9, d["x"], d["y"], p], axis=1).astype(np.float64)   # :35


def read_flo(flow_path):
    with open(flow_path, "rb") as f:                                      # loader_utils.py:55
        magic = np.fromfile(f, np.float32, count=1)
        if 202021.25 != magic:                                            # :57-58 (the reference prints and returns None)
            return None
        w = np.fromfile(f, np.int32, count=1)
        h = np.fromfile(f, np.int32, count=1)
        data = np.fromfile(f, np.float32, count=int(2 * w[0] * h[0]))
        return np.resize(data, (h[0], w[0], 2))                           # :64


def _clamp(i, j, height, width):                                          # HREM.py:30-39
    i = height - 1 if i >= height else (0 if i < 0 else i)
    j = width - 1 if j >= width else (0 if j < 0 else j)
    return i, j


def motion_propagate(fflow, height, width, mesh_size=16, radius=3):
    u, v = fflow[..., 0], fflow[..., 1]                                   # HREM.py:46-47
    mesh_cols, mesh_rows = width // mesh_size, height // mesh_size        # :50
    xm = np.zeros((mesh_size, mesh_size), dtype=float)
    ym = np.zeros((mesh_size, mesh_size), dtype=float)
    for i in range(mesh_size):
        for j in range(mesh_size):
            xs, ys = [], []
            for r in range(radius):                                       # :58-81: four mirrored samples per radius
                ox, oy = r * mesh_rows // 2, r * mesh_cols // 2
                for si, sj in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
                    pi, pj = _clamp(mesh_rows * i + si * ox, mesh_cols * j + sj * oy, height, width)
                    xs.append(u[pi, pj])
                    ys.append(v[pi, pj])
            xs.sort(); ys.sort()                                          # :86-92: upper median of the 12 samples
            xm[i, j] = xs[len(xs) // 2]
            ym[i, j] = ys[len(ys) // 2]
    pad = 2                                                               # :95-101: replicate border, 5x5 median, crop
    xf = medfilt2d(np.pad(xm, pad, mode="edge"), [5, 5])
    yf = medfilt2d(np.pad(ym, pad, mode="edge"), [5, 5])
    return xf[pad:pad + mesh_size, pad:pad + mesh_size], yf[pad:pad + mesh_size, pad:pad + mesh_size]


def flow_error(flow_gt, flow_pred, event_img=None, is_car=False, evaluation_type="dense"):
    """flow_*: (2,H,W) float arrays.  Returns the reference's 7-tuple
    (AEE, percent_1_AEE, percent_3_AEE, n_points, AEE_sum, AEE_gt, AEE_gt_sum) - test_mvsec.py:291-346."""
    gt = np.transpose(flow_gt, (1, 2, 0))
    pr = np.transpose(flow_pred, (1, 2, 0))
    max_row = 190 if is_car else gt.shape[1]                              # :296-298 (shape[1] is the WIDTH: a reference quirk)
    gt, pr = gt[:max_row, :], pr[:max_row, :]
    mask = (~np.isinf(gt[:, :, 0])) & (~np.isinf(gt[:, :, 1])) & (np.linalg.norm(gt, axis=2) > 0)
    if evaluation_type == "sparse":
        mask = mask & (np.squeeze(event_img)[:max_row, :] > 0)            # :304-308
    g, p = gt[mask, :], pr[mask, :]
    ee = np.linalg.norm(g - p, axis=-1)
    ee_gt = np.linalg.norm(g, axis=-1)
    n = ee.shape[0]
    p1 = float((ee < 1.0).sum() / float(n + 1e-5))                        # :322-323
    p3 = float(((ee < 3.0) | (ee < 0.1 * ee_gt)).sum()) / float(n + 1e-5)  # :327
    if ee.sum() == 0:                                                     # :332-338
        return 0.0, p1, p3, n, 0.0, 0.0, 0.0
    return float(ee.mean()), p1, p3, n, float(ee.sum()), float(ee_gt.mean()), float(ee_gt.sum())
