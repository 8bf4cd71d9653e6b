"""Dataset front-end rows and A15: the oracle against goldens produced by executing the reference's
own function sources (tests/golden/make_golden_data.py), and the product's host code against the oracle.  CPU only."""
import os

import numpy as np
import pytest
import torch

from eemflow_amd import hrem
from oracle import data_oracle as D


def test_event_npz_reader(golden, tmp_path):
    g = golden("data_rows.npz")
    ev = hrem.synthetic_hrem_events(int(g["ev_seed"]), int(g["ev_n"]), 720, 1280)
    p = os.path.join(tmp_path, "events1.npz")
    hrem.write_events_npz(p, ev)
    assert np.array_equal(D.get_compressed_events(p), g["ev_ref"])           # oracle == reference
    got = hrem.get_compressed_events(p)
    assert got.dtype == np.float64 and np.array_equal(got, g["ev_ref"])      # product == reference
    assert set(np.unique(got[:, 3])) <= {-1.0, 1.0}


def test_flo_reader(golden, tmp_path):
    g = golden("data_rows.npz")
    h, w = g["flo_hw"]
    fl = hrem.synthetic_flow(int(g["flo_seed"]), h, w)
    p = os.path.join(tmp_path, "flow.flo")
    hrem.write_flo(p, fl)
    assert np.array_equal(D.read_flo(p), g["flo_ref"]) and np.array_equal(hrem.read_flo(p), g["flo_ref"])
    assert np.array_equal(g["flo_ref"], fl)                                   # round trip
    with open(p, "r+b") as f:
        f.write(b"\0\0\0\0")
    assert hrem.read_flo(p) is None and D.read_flo(p) is None                 # wrong magic: the reference returns None


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_motion_propagate(golden, tag):
    g = golden("data_rows.npz")
    h, w = g[f"mp_{tag}_hw"]
    f = hrem.synthetic_flow(int(g[f"mp_{tag}_seed"]), h, w)
    ox, oy = D.motion_propagate(f, h, w)
    assert np.array_equal(ox, g[f"mp_{tag}_x"]) and np.array_equal(oy, g[f"mp_{tag}_y"])     # oracle == reference, bit-exact
    px, py = hrem.motion_propagate(f, h, w)
    assert px.shape == (16, 16) and np.array_equal(px, ox) and np.array_equal(py, oy)       # vectorised product == oracle


def test_flow_error_oracle(golden):
    g = golden("data_rows.npz")
    h, w = g["fe_hw"]
    gt, pred, ev = hrem.flow_error_inputs(int(g["fe_seed"]), h, w)
    i = 0
    for et in ("dense", "sparse"):
        for car in (False, True):
            np.testing.assert_allclose(D.flow_error(gt, pred, ev, is_car=car, evaluation_type=et), g["fe_cases"][i], rtol=1e-6)
            i += 1
    gz = np.nan_to_num(gt, posinf=1.0)
    np.testing.assert_allclose(D.flow_error(gz, gz.copy(), ev), g["fe_cases"][4], rtol=1e-12)


def test_dataset_needs_gpu_and_layout(tmp_path):
    """Directory walk of HREM.py:154-190 (no GPU needed to list samples)."""
    for split, sub in (("train", "dt1/s0"), ("train", "dt1/s1"), ("test", "dt1/seqA/s0")):
        d = os.path.join(tmp_path, "dataset/HREM", split, sub)
        os.makedirs(d)
        for k in ("events1.npz", "events2.npz"):
            hrem.write_events_npz(os.path.join(d, k), hrem.synthetic_hrem_events(1, 10, 720, 1280))
    os.makedirs(os.path.join(tmp_path, "dataset/HREM/train/dt1/incomplete"))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    tr = hrem.HREMEventFlow(args, train=True, root=str(tmp_path))
    assert len(tr) == 2 and [s["names"] for s in tr.data_ls] == ["s0", "s1"]
    te = hrem.HREMEventFlow(args, train=False, root=str(tmp_path))
    assert list(te.nori_list) == ["seqA"]
    te.change_test_sequence("seqA")
    assert len(te) == 1


def test_mvsec_file_lists_crop_and_event_mask(tmp_path):
    """MVSEC.py:60-93 / :201-227 file lists, CenterCrop offsets, and the event mask against np.histogram2d itself."""
    from eemflow_amd import mvsec
    args = {"eval_type": "sparse", "num_voxel_bins": 5, "sequence": "indoor_flying2"}
    ds = mvsec.MvsecEventFlow(args, train=False, root=str(tmp_path))
    assert len(ds) == 2199 - 314 and ds.names[0] == 314
    assert ds.flow_list[0].endswith("dataset/MVSEC/indoor_flying2/flowgt_dt1/314.npy")
    assert ds.event_list[0].endswith("dataset/MVSEC/indoor_flying2/event/000315.h5")
    assert len(ds.event_list) == len(ds) + 1 and ds.event_list[-1].endswith("002200.h5")
    ds.change_test_sequence("indoor_flying1")
    assert "dataset/MVSEC_test/indoor_flying1/" in ds.flow_list[0] and ds.sequence == "indoor_flying1_new"
    ds.change_test_sequence("outdoor_day1")
    assert len(ds) == 3000 - 245
    d4 = mvsec.MvsecEventFlow_dt4(dict(args, sequence="indoor_flying1"), train=False, root=str(tmp_path))
    assert "dataset/MVSEC/indoor_flying1/flowgt_dt4/314.npy" in d4.flow_list[0]
    assert len(d4.event_list) == len(d4) + 5 and d4.event_list[-1].endswith("002204.h5")
    tr = mvsec.MvsecEventFlow(dict(args, aug_params={"crop_size": [256, 256], "do_flip": True}), train=True, root=str(tmp_path))
    assert tr.dense_augmentor is not None and tr.dense_augmentor.crop_size == [256, 256]
    # torchvision CenterCrop((256,256)) of 260x346: top 2, left 45
    x = torch.arange(260 * 346).view(1, 260, 346)
    c = mvsec.center_crop(x, (256, 256))
    assert tuple(c.shape) == (1, 256, 256) and int(c[0, 0, 0]) == 2 * 346 + 45
    rng = np.random.default_rng(5)
    for frac in (False, True):
        n, h, w = 4000, 260, 346
        xs = rng.integers(-2, w + 3, n).astype(np.float64)
        ys = rng.integers(-2, h + 3, n).astype(np.float64)
        if frac:
            xs, ys = xs + rng.uniform(0, 1, n), ys + rng.uniform(0, 1, n)
        xs[:3], ys[:3] = w, h                                     # the closed right edge of the last bin
        feats = np.stack([np.zeros(n), xs, ys, np.ones(n)], 1)
        hist, _, _ = np.histogram2d(x=xs, y=ys, bins=(w, h), range=[[0, w], [0, h]])
        assert np.array_equal(mvsec.event_mask(feats, h, w), hist.transpose() > 0)
    with pytest.raises(FileNotFoundError):
        mvsec.get_events(os.path.join(tmp_path, "nope.h5"))


def test_augmentors_match_reference_classes(golden):
    """FlowAugmentor (no-resize path) and DenseSparseAugmentor against outputs of the reference's own classes under the same
    numpy seed: bit-identical arrays (flips, flow sign changes, random crop)."""
    from eemflow_amd.augmentor import DenseSparseAugmentor, FlowAugmentor
    g = golden("augmentor.npz")
    for k, (seed, h, w, ch, cw, flip) in enumerate(g["cases"].tolist()):
        rng = np.random.default_rng(100 + seed)
        a, b, da, db = (rng.standard_normal((h, w, 3)).astype(np.float32) for _ in range(4))
        fl = rng.standard_normal((h, w, 2))
        np.random.seed(seed)
        got = FlowAugmentor(crop_size=[ch, cw], do_flip=bool(flip))(a, b, fl, without_resize=True)
        for i, arr in enumerate(got):
            assert arr.flags["C_CONTIGUOUS"] and np.array_equal(arr, g[f"flow_nr_{k}_{i}"]), (k, i)
        np.random.seed(seed)
        got = DenseSparseAugmentor(crop_size=[ch, cw], do_flip=bool(flip))(a, b, da, db, fl)
        for i, arr in enumerate(got):
            assert arr.dtype == g[f"dense_{k}_{i}"].dtype and np.array_equal(arr, g[f"dense_{k}_{i}"]), (k, i)


def test_flow_augmentor_with_rescaling():
    """FlowAugmentor's default path (utils/augumentor.py:158-200: random rescale, flips, random crop).  cv2 is absent, so the resize is
    a restatement of cv2.resize(INTER_LINEAR) for float arrays - parity unpinned against cv2; here it is held to torch's bilinear
    interpolation (align_corners=False: the same half-pixel rule), and the whole transform to its own definition under a fixed seed:
    crop-sized C-contiguous outputs, flow scaled by the drawn factors."""
    import torch
    import torch.nn.functional as F
    from eemflow_amd.augmentor import FlowAugmentor, resize_linear
    rng = np.random.default_rng(7)
    img = rng.standard_normal((37, 53, 3)).astype(np.float32)
    for fx, fy in ((2.0, 2.0), (1.0, 1.0)):                      # src * f an integer: cv2's 1 / f map IS the src / dst map torch uses
        got = resize_linear(img, fx, fy)
        oh, ow = int(round(37 * fy)), int(round(53 * fx))
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == (oh, ow, 3) and got.dtype == np.float32
        assert np.abs(got - ref).max() < 5e-5                      # (torch forms the source index in float32)
    # src * f NOT an integer: the map uses the given factor (cv2: dsize = round(src * f), sample at (d + 0.5) / f - 0.5), written out per pixel
    small = rng.standard_normal((7, 9)).astype(np.float32)
    for fx, fy in ((1.3, 0.8), (0.71, 1.45)):
        got = resize_linear(small, fx, fy)
        oh, ow = int(round(7 * fy)), int(round(9 * fx))
        assert got.shape == (oh, ow)
        for y in range(oh):
            sy = (y + 0.5) / fy - 0.5
            y0 = int(np.floor(sy)); ty = sy - y0
            for x in range(ow):
                sx = (x + 0.5) / fx - 0.5
                x0 = int(np.floor(sx)); tx = sx - x0
                v = lambda yy, xx: float(small[min(max(yy, 0), 6), min(max(xx, 0), 8)])   # noqa: E731
                want = (v(y0, x0) * (1 - tx) + v(y0, x0 + 1) * tx) * (1 - ty) + (v(y0 + 1, x0) * (1 - tx) + v(y0 + 1, x0 + 1) * tx) * ty
                assert abs(float(got[y, x]) - want) < 1e-5, (fx, fy, y, x)
    a, b = (rng.standard_normal((120, 160, 3)).astype(np.float32) for _ in range(2))
    fl = rng.standard_normal((120, 160, 2))
    np.random.seed(3)
    o1, o2, of = FlowAugmentor(crop_size=[64, 96], do_flip=True)(a, b, fl)
    assert o1.shape == (64, 96, 3) and o2.shape == (64, 96, 3) and of.shape == (64, 96, 2)
    assert all(x.flags["C_CONTIGUOUS"] for x in (o1, o2, of))
    np.random.seed(3)                                            # the same draws again: deterministic
    p1, _, pf = FlowAugmentor(crop_size=[64, 96], do_flip=True)(a, b, fl)
    assert np.array_equal(o1, p1) and np.array_equal(of, pf)
