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


def test_evaluation_with_frames_in_flight_equals_the_sequential_loop(tmp_path, monkeypatch):
    """TestRaftEvents.test_multi_sequence(frames_in_flight=3): three replicas on three streams take the samples round robin, the
    statistics are fetched two samples late - the same per-sample numbers in the same order, the same mean AEE, bit for bit when the
    encoder's Winograd form is pinned (EEM_WINO4_LAYERS; by default frames in flight move two layer pairs to F(4x4,3x3), which
    agrees with F(2x2,3x3) to fp32 round-off: the mean AEE then agrees to 1e-5)."""
    from eemflow_amd import EEMFlow
    from eemflow_amd.harness import TestRaftEvents
    from eemflow_amd.weights import seeded_state_dict
    root = str(tmp_path)
    for i in range(7):
        d = os.path.join(root, "dataset/HREM/test/dt1/seqB/%06d" % (i + 1))
        os.makedirs(d)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(40 + i, 30000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(60 + i, 30000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(80 + i, 720, 1280))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    def run_all():
        net = EEMFlow("", 5, 5)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(5).items()})
        net = net.to(DEV)
        out = []
        for nfl, threads in ((1, 0), (3, 0), (3, 2)):
            ev = TestRaftEvents(hrem.HREMEventFlow(args, train=False, root=root), (720, 1280))
            out.append((ev.test_multi_sequence(net, epoch=0, sequence_list=["seqB"], stride=1, frames_in_flight=nfl, loader_threads=threads),
                        list(ev.logger.lines)))
        assert net.frames_in_flight == 1                                       # the hint is restored
        return out
    monkeypatch.setenv("EEM_WINO4_LAYERS", "7")                                # read when a context loads its weights
    out = run_all()
    assert all(o == out[0] for o in out[1:])
    monkeypatch.delenv("EEM_WINO4_LAYERS")
    out = run_all()
    assert out[1] == out[2] and abs(out[0][0] - out[1][0]) < 1e-5


def test_harness_eval_and_train_on_synthetic_hrem(tmp_path):
    """test_multi_sequence / train_iters over a two-sample synthetic HREM tree: runs the whole row chain
    (files -> GPU voxelizer -> model -> flow_error / optimisation step) and the checkpoint writer."""
    from eemflow_amd import EEMFlow
    from eemflow_amd.harness import TestRaftEvents, TrainRaftEvents, load_checkpoint, save_checkpoint
    from eemflow_amd.weights import seeded_state_dict
    root = str(tmp_path)
    for split, sub in (("test", "dt1/seqA/000001"), ("test", "dt1/seqA/000002"), ("train", "dt1/000001"), ("train", "dt1/000002")):
        d = os.path.join(root, "dataset/HREM", split, sub)
        os.makedirs(d)
        seed = hash(sub) % 1000
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(seed, 20000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(seed + 1, 20000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(seed + 2, 720, 1280))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(3).items()})
    net = net.to(DEV)
    # ---- evaluation: mean AEE equals the oracle's flow_error on the same prediction
    ds = hrem.HREMEventFlow(args, train=False, root=root)
    ev = TestRaftEvents(ds, (720, 1280))
    mean_aee = ev.test_multi_sequence(net, epoch=0, sequence_list=["seqA"], stride=1)
    ds.change_test_sequence("seqA")
    ref = []
    with torch.no_grad():
        for i in range(2):
            s = ds[i]
            pred = net(s["event_volume_old"][None], s["event_volume_new"][None])[1][-1]
            ref.append(D.flow_error(s["flow"].cpu().numpy(), pred[0].cpu().numpy())[0])
    assert abs(mean_aee - float(np.mean(ref))) < 1e-4
    assert any(line.startswith("seqA: Mean AEE") for line in ev.logger.lines)
    # ---- training on the 16x16 mesh-flow target (out_mesh_size=True, EEMFlow.py:126-132)
    tnet = EEMFlow("", 5, 5, out_mesh_size=True)
    tnet.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(3).items()})
    tnet = tnet.to(DEV)
    tds = hrem.HREMEventFlow(args, train=True, root=root)
    loader = torch.utils.data.DataLoader(tds, batch_size=2, shuffle=False, num_workers=0)
    tr = TrainRaftEvents(loader, (720, 1280), lr=1e-4, num_steps=100)
    w0 = tnet.state_dict()["out_conv.weight"].clone()
    tr.train_iters(tnet, start_epoch=0, val_iters=1)
    p = os.path.join(root, "lasted_ckpt.pth.tar")
    save_checkpoint(p, tnet, epoch=0, trainer=tr.trainer)
    fresh = EEMFlow("", 5, 5, out_mesh_size=True)
    assert load_checkpoint(p, fresh) == 0
    assert not torch.equal(fresh.state_dict()["out_conv.weight"], w0.cpu())          # the step changed the weights


def _mvsec_tree(root, seq, frames, dt4=False):
    """Synthetic MVSEC tree: per-frame event tables exported as npz (ts seconds, x, y, p in {-1,+1}), flow as .npy."""
    from eemflow_amd.hrem import synthetic_flow, synthetic_hrem_events
    ev_dir = os.path.join(root, "dataset/MVSEC", seq, "event")
    fl_dir = os.path.join(root, "dataset/MVSEC", seq, "flowgt_dt4" if dt4 else "flowgt_dt1")
    os.makedirs(ev_dir, exist_ok=True)
    os.makedirs(fl_dir, exist_ok=True)
    events = {}
    for f in range(frames[0] + 1, frames[1] + 8):
        ev = synthetic_hrem_events(100 + f, 3000 + 10 * f, 260, 346, t_span=0.02)
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        ev[:, 0] += 0.02 * f                                        # consecutive frames, increasing time
        events[f] = ev
        np.savez(os.path.join(ev_dir, "%06d.npz" % f), ts=ev[:, 0], x=ev[:, 1], y=ev[:, 2], p=ev[:, 3])
    for i in range(*frames):
        fl = synthetic_flow(200 + i, 260, 346)                      # (H,W,2): the loader transposes
        np.save(os.path.join(fl_dir, "%d.npy" % i), fl)
    return events


@pytest.mark.parametrize("dt4", [False, True])
def test_mvsec_dataset_end_to_end(tmp_path, dt4):
    from eemflow_amd import mvsec
    root = str(tmp_path)
    frames = (10, 13)
    events = _mvsec_tree(root, "indoor_flying2", frames, dt4)
    cls = mvsec.MvsecEventFlow_dt4 if dt4 else mvsec.MvsecEventFlow
    args = {"eval_type": "sparse", "num_voxel_bins": 5, "sequence": "indoor_flying2"}
    ds = cls(args, train=False, root=root, valid_time_index={"indoor_flying2": [frames]})
    assert len(ds) == 3
    k = 4 if dt4 else 1
    for idx in (0, 2):
        s = ds[idx]
        raw = ds.get_sample(idx)
        old = np.concatenate([events[frames[0] + idx + 1 + i] for i in range(k)])
        new = np.concatenate([events[frames[0] + idx + 2 + i] for i in range(k)])
        for key, ev in (("event_volume_old", old), ("event_volume_new", new)):
            feats = O.event_sequence(ev, 1e6, True)
            ref = O.voxelize(feats, 5, 260, 346, normalize=True)
            assert raw[key].is_cuda and float((raw[key].cpu() - torch.from_numpy(ref)).abs().max()) < 1e-4
            assert torch.equal(s[key].cpu(), torch.from_numpy(ref[:, 2:258, 45:301]).to(s[key].dtype)) or \
                float((s[key].cpu() - torch.from_numpy(ref[:, 2:258, 45:301])).abs().max()) < 1e-4
        hist, _, _ = np.histogram2d(x=old[:, 1], y=old[:, 2], bins=(346, 260), range=[[0, 346], [0, 260]])
        assert np.array_equal(raw["event_valid"][0].numpy(), hist.transpose() > 0)
        assert tuple(s["event_valid"].shape) == (1, 256, 256) and tuple(s["flow"].shape) == (2, 256, 256)
        fl = synthetic_flow_chw(200 + frames[0] + idx)
        assert torch.equal(s["flow"], torch.from_numpy(fl[:, 2:258, 45:301]))
        assert s["valid"].dtype == torch.bool and s["idx"] == frames[0] + idx
    tr = cls(args, train=True, root=root, valid_time_index={"indoor_flying2": [frames]})
    t = tr[1]
    assert tuple(t["event_volume_old"].shape) == (5, 260, 346) and tuple(t["valid"].shape) == (260, 346)
    # with aug_params: DenseSparseAugmentor (flips + 256x256 random crop), reproducible under the numpy seed
    ta = cls(dict(args, aug_params={"crop_size": [256, 256], "do_flip": True}), train=True, root=root,
             valid_time_index={"indoor_flying2": [frames]})
    np.random.seed(7)
    s1 = ta[1]
    np.random.seed(7)
    s2 = ta[1]
    assert tuple(s1["event_volume_old"].shape) == (5, 256, 256) and tuple(s1["flow"].shape) == (2, 256, 256)
    assert torch.equal(s1["event_volume_new"], s2["event_volume_new"]) and torch.equal(s1["flow"], s2["flow"])
    assert torch.equal(s1["event_volume_old"], s1["d_event_volume_old"]) and tuple(s1["valid"].shape) == (256, 256)


def synthetic_flow_chw(seed):
    from eemflow_amd.hrem import synthetic_flow
    return np.ascontiguousarray(synthetic_flow(seed, 260, 346).transpose(2, 0, 1))


def test_harness_evaluates_mvsec_sparse(tmp_path):
    """test_multi_sequence over a synthetic MVSEC tree with the 'sparse' metric (event mask -> flow_error on the GPU)."""
    from eemflow_amd import EEMFlow, mvsec
    from eemflow_amd.harness import TestRaftEvents
    from eemflow_amd.weights import seeded_state_dict
    root = str(tmp_path)
    frames = (20, 23)
    _mvsec_tree(root, "indoor_flying2", frames)
    ds = mvsec.MvsecEventFlow({"eval_type": "sparse", "num_voxel_bins": 5, "sequence": "indoor_flying2"}, train=False, root=root,
                              valid_time_index={"indoor_flying2": [frames]})
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(3).items()})
    net = net.to(DEV)
    aee = TestRaftEvents(ds, (256, 256)).test_multi_sequence(net, sequence_list=["indoor_flying2"], stride=1)
    assert np.isfinite(aee) and aee > 0


def test_cli_train_then_test_on_synthetic_hrem(tmp_path):
    """`python -m eemflow_amd.cli train / test` with the reference scripts' flags on a two-sample synthetic HREM tree: run folder,
    config.json, train.log, lasted_ckpt.pth.tar (reference layout), then evaluation from that checkpoint."""
    from eemflow_amd import cli
    root = str(tmp_path)
    for split, sub in (("test", "dt1/seqA/000001"), ("train", "dt1/000001"), ("train", "dt1/000002")):
        d = os.path.join(root, "dataset/HREM", split, sub)
        os.makedirs(d)
        seed = 7 + len(sub)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(seed, 20000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(seed + 1, 20000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(seed + 2, 720, 1280))
    common = ["--data_root", root, "--save_root", root, "--lr", "1e-4", "--wd", "1e-5"]
    torch.manual_seed(11)                                         # the model's init draws from torch's generator
    run = cli.main(["train", *common, "-bs", "2", "--train_iters", "2", "--val_iters", "1"])
    assert run.endswith("exp_HREM_meshflow/EEMFlow_dt1/lr0.000100_we0.000010")
    for f in ("config.json", "train.log", "lasted_ckpt.pth.tar"):
        assert os.path.exists(os.path.join(run, f)), f
    ck = torch.load(os.path.join(run, "lasted_ckpt.pth.tar"), weights_only=False)
    assert ck["epoch"] == 1 and len(ck["state_dict"]) == 66 and "iteration" in ck
    aee = cli.main(["test", *common, "--checkpoint", os.path.join(run, "lasted_ckpt.pth.tar")])
    assert np.isfinite(aee) and os.path.exists(os.path.join(root, "HREM_testset/EEMFlow_dt1/test.log"))
    # the same evaluation with samples in flight and loader threads, the same training with `-n` threads feeding it
    # (frames in flight move two encoder layer pairs to the F(4x4,3x3) Winograd form: the same flow to fp32 round-off)
    assert abs(cli.main(["test", *common, "--checkpoint", os.path.join(run, "lasted_ckpt.pth.tar"), "--frames_in_flight", "3",
                         "--loader_threads", "2"]) - aee) < 1e-5
    first = torch.load(os.path.join(run, "lasted_ckpt.pth.tar"), weights_only=False)["state_dict"]
    torch.manual_seed(11)
    cli.main(["train", *common, "-bs", "2", "--train_iters", "2", "--val_iters", "1", "-n", "2"])
    again = torch.load(os.path.join(run, "lasted_ckpt.pth.tar"), weights_only=False)["state_dict"]
    # two samples = one batch per epoch, in either order: the same two steps up to summation order (Adam's first steps move a
    # weight by ~lr whatever its gradient's size, so an element whose gradient is ~eps may differ by 2 lr)
    diff = torch.cat([(first[k] - again[k]).abs().reshape(-1) for k in first])
    assert float(diff.max()) < 3e-4 and float(diff.median()) < 1e-6


def test_coalesced_evaluation_agrees_with_the_one_sample_loop(tmp_path, monkeypatch):
    """TestRaftEvents.test_multi_sequence(coalesce=4): four samples per voxelizer launch sequence and per EEMFlow.forward_many call, raw
    volumes normalised by pconv1_1 (HREMEventFlow(deferred_norm=True)) - the same lines in the same order; per-sample AEE within 1e-4 of
    the one-sample loop (another Winograd form in the batched chain, one multiply per voxel in the normalisation); seven samples: a last
    chunk of three.  The dataset's deferred samples carry the event_valid of the normalised volume."""
    from eemflow_amd import EEMFlow
    from eemflow_amd.harness import TestRaftEvents
    from eemflow_amd.weights import seeded_state_dict
    root = str(tmp_path)
    for i in range(7):
        d = os.path.join(root, "dataset/HREM/test/dt1/seqB/%06d" % (i + 1))
        os.makedirs(d)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(40 + i, 30000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(60 + i, 30000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(80 + i, 720, 1280))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(5).items()})
    net = net.to(DEV)
    res = []
    for co, nfl, deferred in ((1, 1, False), (4, 2, True), (4, 1, False), (16, 2, True)):
        ds = hrem.HREMEventFlow(args, train=False, root=root, deferred_norm=deferred)
        ds.change_test_sequence("seqB")
        ev = TestRaftEvents(ds, (720, 1280))
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mean = ev.test_multi_sequence(net, epoch=0, sequence_list=["seqB"], stride=1, frames_in_flight=nfl, coalesce=co)
        per = [float(ln.split("AEE:")[1].split()[0]) for ln in buf.getvalue().splitlines() if " / " in ln and "AEE:" in ln]
        ids = [ln.split("/")[0].strip() for ln in buf.getvalue().splitlines() if " / " in ln and "AEE:" in ln]
        res.append((mean, per, ids))
    assert all(r[2] == res[0][2] and len(r[1]) == 7 for r in res)
    for r in res[1:]:
        assert abs(r[0] - res[0][0]) < 1e-4
        assert max(abs(a - b) for a, b in zip(r[1], res[0][1])) < 1e-4
    # event_valid of a deferred sample = the bin sum of the normalised volume
    a = hrem.HREMEventFlow(args, train=False, root=root)
    b = hrem.HREMEventFlow(args, train=False, root=root, deferred_norm=True)
    a.change_test_sequence("seqB"); b.change_test_sequence("seqB")
    sa, sb = a[2], b.get_samples([2, 3])[0]
    assert sb["deferred_norm"] and float((sa["event_valid"] - sb["event_valid"]).abs().max()) < 1e-4
    assert torch.equal(sa["flow"], sb["flow"])


def test_flow_error_many_equals_the_single_calls():
    """eemflow_flow_error_many: the samples of a coalesced call scored by one launch - per sample the five sums of eemflow_flow_error
    (test_mvsec.py:291-346) to f64 round-off (the atomics of a row commute up to rounding), dense and sparse, is_car rows."""
    from eemflow_amd.metrics import flow_error_sums, flow_error_sums_many
    g = torch.Generator().manual_seed(3)
    n, h, w = 5, 260, 348
    gts = [torch.randn(1, 2, h, w, generator=g).to(DEV) * 3 for _ in range(n)]
    prs = [(gts[i] + torch.randn(1, 2, h, w, generator=g).to(DEV)) for i in range(n)]
    evs = [(torch.rand(h, w, generator=g) < 0.3).float().to(DEV) for _ in range(n)]
    gts[1][0, :, :7, :9] = float("inf")
    gts[2][0, :, 100:120] = 0.0
    for kind, car in (("dense", False), ("sparse", False), ("dense", True)):
        many = flow_error_sums_many(gts, prs, evs if kind == "sparse" else None, is_car=car, evaluation_type=kind).cpu()
        for i in range(n):
            one = flow_error_sums(gts[i], prs[i], evs[i] if kind == "sparse" else None, is_car=car, evaluation_type=kind).cpu()
            assert torch.allclose(many[i], one, rtol=1e-12, atol=0), (kind, car, i)
            assert float(many[i][2]) == float(one[2])                     # the counts exactly
    with pytest.raises(ValueError):
        flow_error_sums_many(gts * 4, prs * 4)


def test_coalesced_evaluation_of_the_other_models(tmp_path):
    """test_multi_sequence(coalesce=3) with EEMFlow_cdc (eemplus_forward_many behind EEMFlow_cdc.forward_many): the same samples in the
    same order as the one-sample loop; the batched chain picks other tiles on the coarse levels, and past the first warped level the
    reference's own `>= 1.0` mask moves flows by ~0.1 px on round-off (tests/test_gpu_plus.py's header) - the per-sample AEE agrees to
    that sensitivity."""
    from eemflow_amd.eemflow_plus import EEMFlow_cdc
    from eemflow_amd.harness import TestRaftEvents
    from eemflow_amd.plus_weights import seeded_from_shapes
    import contextlib
    import io
    root = str(tmp_path)
    for i in range(5):
        d = os.path.join(root, "dataset/HREM/test/dt1/seqC/%06d" % (i + 1))
        os.makedirs(d)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(140 + i, 30000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(160 + i, 30000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(180 + i, 720, 1280))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    net = EEMFlow_cdc("", 3, 5).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 3).items()})
    net = net.to(DEV)
    res = []
    for co in (1, 3):
        ds = hrem.HREMEventFlow(args, train=False, root=root)
        ds.change_test_sequence("seqC")
        ev = TestRaftEvents(ds, (720, 1280))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mean = ev.test_multi_sequence(net, epoch=0, sequence_list=["seqC"], stride=1, coalesce=co)
        lines = [ln for ln in buf.getvalue().splitlines() if " / " in ln and "AEE:" in ln]
        res.append((mean, [float(ln.split("AEE:")[1].split()[0]) for ln in lines], [ln.split("/")[0].strip() for ln in lines]))
    assert res[0][2] == res[1][2] and len(res[1][1]) == 5
    assert max(abs(a - b) for a, b in zip(res[0][1], res[1][1])) < 5e-2 and abs(res[0][0] - res[1][0]) < 5e-2


def test_evaluation_loop_asks_eraft_for_the_last_prediction_only(tmp_path):
    """test_multi_sequence with ERAFT: the loop reads preds[-1] (test_mvsec.py:1455), so the harness switches ERAFT.final_only on for the
    call and back afterwards - the same AEE lines as with the full list (the last prediction is bit for bit the same), one sample at a
    time and three per forward_many call."""
    from eemflow_amd.eraft import ERAFT
    from eemflow_amd.eraft_weights import seeded_from_shapes
    from eemflow_amd.harness import TestRaftEvents
    import contextlib
    import io
    root = str(tmp_path)
    for i in range(3):
        d = os.path.join(root, "dataset/HREM/test/dt1/seqE/%06d" % (i + 1))
        os.makedirs(d)
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(240 + i, 30000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(260 + i, 30000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(280 + i, 720, 1280))
    args = {"eval_type": "dense", "event_interval": "dt1", "num_voxel_bins": 5}
    net = ERAFT("", 5).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_from_shapes({k: tuple(v.shape) for k, v in net.state_dict().items()}, 4).items()})
    net = net.to(DEV)
    seen = []
    orig = ERAFT.forward

    def spy(self, *a, **kw):
        seen.append(self.final_only)
        return orig(self, *a, **kw)

    res = []
    for co, patched in ((1, True), (1, False), (3, False)):
        ds = hrem.HREMEventFlow(args, train=False, root=root)
        ds.change_test_sequence("seqE")
        ev = TestRaftEvents(ds, (720, 1280))
        buf = io.StringIO()
        if patched:
            ERAFT.forward = spy
        try:
            with contextlib.redirect_stdout(buf):
                mean = ev.test_multi_sequence(net, epoch=0, sequence_list=["seqE"], stride=1, coalesce=co)
        finally:
            ERAFT.forward = orig
        res.append((mean, [float(ln.split("AEE:")[1].split()[0]) for ln in buf.getvalue().splitlines() if " / " in ln and "AEE:" in ln]))
    assert seen and all(seen) and net.final_only is False
    # the full list's last prediction, scored by hand
    ds = hrem.HREMEventFlow(args, train=False, root=root)
    ds.change_test_sequence("seqE")
    s0 = ds[0]
    with torch.no_grad():
        full = net(s0["event_volume_old"].to(DEV)[None].float(), s0["event_volume_new"].to(DEV)[None].float())[1]
    assert len(full) == 12
    epe = float((full[-1][0] - s0["flow"].to(DEV)).pow(2).sum(0).sqrt().mean())
    assert abs(epe - res[0][1][0]) < 1e-4 * max(1.0, epe)
    assert len(res[0][1]) == 3 and res[0][1] == res[1][1]
    assert max(abs(a - b) for a, b in zip(res[2][1], res[0][1])) < 2e-3 * max(1.0, max(res[0][1]))
