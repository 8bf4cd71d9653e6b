"""GPU side of the dataset / metrics rows: flow_error on the device against the reference-generated golden, and the
HREM dataset end to end on synthetic files (npz + .flo -> mesh flow, GPU-voxelized event volumes).  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

from eemflow_amd import hrem
from eemflow_amd.metrics import flow_error
from eemflow_amd.voxelizer import EventSequence
from oracle import data_oracle as D
from oracle import eemflow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_flow_error_vs_reference_golden(golden):
    g = golden("data_rows.npz")
    h, w = g["fe_hw"]
    gt, pred, ev = hrem.flow_error_inputs(int(g["fe_seed"]), h, w)
    tg, tp, te = (torch.from_numpy(a).to(DEV) for a in (gt[None], pred[None], ev))
    i = 0
    for et in ("dense", "sparse"):
        for car in (False, True):
            got = flow_error(tg, tp, te, is_car=car, evaluation_type=et)
            ref = g["fe_cases"][i]
            assert got[3] == int(ref[3])                                     # n_points: exact
            np.testing.assert_allclose(got[1:3], ref[1:3], rtol=1e-6)        # percentages: counts are exact
            np.testing.assert_allclose([got[0], got[4], got[5], got[6]], ref[[0, 4, 5, 6]], rtol=2e-6)
            i += 1
    gz = torch.from_numpy(np.nan_to_num(gt, posinf=1.0)[None]).to(DEV)
    assert flow_error(gz, gz.clone(), te) == tuple(g["fe_cases"][4][:3]) + (int(g["fe_cases"][4][3]), 0.0, 0.0, 0.0)


def test_flow_error_full_size_and_errors():
    h, w = 720, 1280
    gt, pred, ev = hrem.flow_error_inputs(31, h, w)
    got = flow_error(torch.from_numpy(gt[None]).to(DEV), torch.from_numpy(pred[None]).to(DEV), torch.from_numpy(ev).to(DEV),
                     evaluation_type="sparse")
    ref = D.flow_error(gt, pred, ev, evaluation_type="sparse")
    assert got[3] == ref[3]
    np.testing.assert_allclose(got[:3], ref[:3], rtol=2e-6)
    with pytest.raises(Exception):
        flow_error(torch.from_numpy(gt[None]), torch.from_numpy(pred[None]))          # CPU tensors: no fallback


def test_hrem_dataset_end_to_end(tmp_path):
    root = str(tmp_path)
    d = os.path.join(root, "dataset/HREM/test/dt1/seq0/000001")
    os.makedirs(d)
    ev1 = hrem.synthetic_hrem_events(41, 20000, 720, 1280)
    ev2 = hrem.synthetic_hrem_events(42, 30000, 720, 1280)
    fl = hrem.synthetic_flow(43, 720, 1280)
    hrem.write_events_npz(os.path.join(d, "events1.npz"), ev1)
    hrem.write_events_npz(os.path.join(d, "events2.npz"), ev2)
    hrem.write_flo(os.path.join(d, "flow.flo"), fl)
    ds = hrem.HREMEventFlow({"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}, train=False, root=root)
    ds.change_test_sequence("seq0")
    s = ds[0]
    assert s["names"] == "000001" and s["event_volume_old"].is_cuda and tuple(s["event_volume_old"].shape) == (5, 720, 1280)
    # event volumes == the voxelizer oracle on the events the reference reader returns
    for key, ev in (("event_volume_old", ev1), ("event_volume_new", ev2)):
        feats = D.get_compressed_events(os.path.join(d, "events1.npz" if key.endswith("old") else "events2.npz"))
        seq = EventSequence(None, {"height": 720, "width": 1280}, features=feats, timestamp_multiplier=1e6, convert_to_relative=True)
        ref = O.voxelize(seq.features, 5, 720, 1280, normalize=True)
        assert float((s[key].cpu() - torch.from_numpy(ref)).abs().max()) < 1e-4
    assert torch.equal(s["event_valid"][0], s["event_volume_old"].sum(0))
    # ground truth: mesh flow (bit-exact vs the oracle) upsampled to full resolution, valid mask
    mx, my = D.motion_propagate(fl, 720, 1280)
    mesh = torch.from_numpy(np.stack([mx, my])).float()
    up = torch.nn.functional.interpolate(mesh[None], size=(720, 1280), mode="bilinear", align_corners=False)[0]
    assert float((s["flow"].cpu() - up).abs().max()) < 1e-5
    assert tuple(s["valid"].shape) == (720, 1280) and float(s["valid"].min()) >= 0
    assert torch.equal(s["fflow"], torch.from_numpy(fl.transpose(2, 0, 1).copy()))
