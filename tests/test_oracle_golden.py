"""The oracle (oracle/eemflow_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from eemflow_amd.weights import CORR_TAPS_53, seeded_state_dict, synthetic_voxel_pair
from oracle import eemflow_oracle as O


def test_pad_table(golden):
    g = golden("pad.npz")
    for h, w, rate, mode, *pad in g["table"].tolist():
        assert O.input_padder_pad(h, w, "chairs" if mode == 0 else "sintel", rate) == pad
    x = torch.from_numpy(g["x"])
    xp = O.replicate_pad(x, g["x_pad"].tolist())
    assert np.array_equal(xp.numpy(), g["x_padded"])
    assert np.array_equal(O.unpad(xp, g["x_pad"].tolist()).numpy(), g["x_unpadded"])


def test_known_pads():
    # SURVEY.md Appendix B (probe-verified on the reference)
    assert O.input_padder_pad(260, 346) == [19, 19, 0, 60]
    assert O.input_padder_pad(720, 1280) == [0, 0, 0, 48]
    assert O.input_padder_pad(512, 960) == [0, 0, 0, 0]


def test_local_corr(golden):
    g = golden("local_corr.npz")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    np.testing.assert_allclose(O.local_corr81(x, y).numpy(), g["cv81"], atol=1e-6)
    np.testing.assert_allclose(O.local_corr53(x, y).numpy(), g["cv53"], atol=1e-6)
    assert tuple(O.CORR_TAPS_53) == tuple(CORR_TAPS_53) and len(CORR_TAPS_53) == 53


def test_decoder(golden):
    g = golden("decoder.npz")
    sd = O.to_torch_sd(seeded_state_dict(int(g["seed"])))
    x = torch.from_numpy(g["x"])
    perm = O.channel_shuffle(torch.arange(100.0).view(1, 100, 1, 1), 5).flatten().numpy().astype(np.int64)
    assert np.array_equal(perm, g["shuffle_perm"])
    np.testing.assert_allclose(O.decoder(sd, "decoder_2.", x).numpy(), g["y"], atol=1e-6)


@pytest.mark.parametrize("tag", ["128x192", "260x346", "100x150"])
def test_eemflow_forward(golden, tag):
    g = golden(f"eemflow_fwd_{tag}.npz")
    h, w = g["hw"].tolist()
    sd = O.to_torch_sd(seeded_state_dict(int(g["seed"])))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w))
    with torch.no_grad():
        flow, st = O.eemflow_forward(sd, e1, e2, keep=True)
    assert st["pad"] == g["pad"].tolist()
    for k in g.files:
        if k in st and k != "pad":
            np.testing.assert_allclose(st[k].numpy(), g[k], atol=2e-6, rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(flow.numpy(), g["flow"], atol=2e-6, rtol=1e-5)


VOX_CASES = ["n20k", "n20k_pol01", "n1", "n2_dt0", "n3_dt0", "n500_raw", "n300_bins3", "n400_unsorted", "n64_const"]


@pytest.mark.parametrize("name", VOX_CASES)
def test_voxelizer(golden, name):
    g = golden("voxel.npz")
    h, w, bins, norm = g[f"{name}_hwb"].tolist()
    feats = O.event_sequence(g[f"{name}_events"], 1e6, True)
    il, _, ir, _ = O.voxel_indices(feats, bins, h, w)
    assert np.array_equal(il, g[f"{name}_idx_left"])        # integer indices: bit-exact
    assert np.array_equal(ir, g[f"{name}_idx_right"])
    grid = O.voxelize(feats, bins, h, w, bool(norm))
    ref = g[f"{name}_grid"]
    assert np.array_equal(np.isnan(grid), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(grid), np.nan_to_num(ref), atol=1e-6, rtol=1e-6)
    if not norm:
        assert np.array_equal(grid, ref)                    # same accumulation order => same bits


# ----------------------------------------------------------------------------- E-RAFT
from oracle import eraft_oracle as R   # noqa: E402


def eraft_sd(seed):
    from eemflow_amd.eraft import ERAFT
    from eemflow_amd.eraft_weights import seeded_from_shapes
    shapes = {k: tuple(v.shape) for k, v in ERAFT("", 5).state_dict().items()}
    return O.to_torch_sd(seeded_from_shapes(shapes, seed))


def test_eraft_lookup_and_pyramid(golden):
    g = golden("eraft_lookup.npz")
    pyr = R.corr_pyramid(torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"]))
    for i, p in enumerate(pyr):
        np.testing.assert_allclose(p.numpy(), g[f"pyr{i}"], atol=1e-6)
    out = R.corr_lookup(pyr, torch.from_numpy(g["coords"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-6)


def test_eraft_convex_upsample(golden):
    g = golden("eraft_upsample.npz")
    up = R.convex_upsample(torch.from_numpy(g["flow"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(up.numpy(), g["up"], atol=1e-6)


@pytest.mark.parametrize("tag", ["128x160", "136x200"])
def test_eraft_forward(golden, tag):
    g = golden(f"eraft_fwd_{tag}.npz")
    h, w = g["hw"].tolist()
    sd = eraft_sd(int(g["seed"]))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w))
    with torch.no_grad():
        preds, st = R.eraft_forward(sd, e1, e2, iters=int(g["iters"]), keep=True)
    assert st["pad"] == g["pad"].tolist()
    for k in ("fmap1", "fmap2", "net0", "inp", "corr0", "net1", "mask1", "delta1"):
        if k in g.files:
            np.testing.assert_allclose(st[k].numpy(), g[k], atol=2e-5, rtol=1e-5, err_msg=k)
    if "pyr1" in g.files:
        np.testing.assert_allclose(st["pyr"][1].numpy(), g["pyr1"], atol=1e-5)
        np.testing.assert_allclose(st["pyr"][3].numpy(), g["pyr3"], atol=1e-5)
    assert np.isfinite(g["preds"]).all()
    np.testing.assert_allclose(torch.stack(preds).numpy(), g["preds"], atol=1e-4, rtol=1e-5)


# ----------------------------------------------------------------------------- training step (A14)
from oracle import train_oracle as T   # noqa: E402


def test_train_loss_and_grads(golden):
    from eemflow_amd.weights import synthetic_gt
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b = int(g["batch"])
    sd = O.to_torch_sd(seeded_state_dict(int(g["seed"])))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(int(g["gt_seed"]), b, h, w))
    loss, metrics, grads, flow = T.loss_and_grads(sd, e1, e2, gt, valid)
    assert abs(loss - float(g["loss"])) < 1e-6 and abs(metrics["epe"] - float(g["epe"])) < 1e-5
    np.testing.assert_allclose(flow.numpy(), g["flow"], atol=2e-6)
    assert list(grads.keys()) == g["grad_keys"].tolist()
    norms = np.array([float(v.double().norm()) for v in grads.values()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    for k in g.files:
        if k.startswith("g:"):
            np.testing.assert_allclose(grads[k[2:]].numpy(), g[k], rtol=1e-3, atol=2e-6, err_msg=k)


def test_train_three_steps(golden):
    from eemflow_amd.weights import synthetic_gt
    g = golden("train_step.npz")
    h, w = g["hw"].tolist()
    b, seed = int(g["batch"]), int(g["seed"])
    sd = O.to_torch_sd(seeded_state_dict(seed))
    batches = []
    for step in range(3):
        e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(seed + 3100 + step, b, h, w))
        gt, valid = (torch.from_numpy(a) for a in synthetic_gt(seed + 3200 + step, b, h, w))
        batches.append((e1, e2, gt, valid))
    losses, lrs, final = T.train_steps(sd, batches, lr=1e-3, num_steps=20)
    np.testing.assert_allclose(losses, g["step_losses"], rtol=1e-5)
    np.testing.assert_allclose(lrs, g["step_lrs"], rtol=1e-7)
    np.testing.assert_allclose([float(v.double().norm()) for v in final.values()], g["final_norms"], rtol=1e-5)
    for k in g.files:
        if k.startswith("p3:"):
            np.testing.assert_allclose(final[k[3:]].numpy(), g[k], atol=2e-5, err_msg=k)


# ----------------------------------------------------------------------------- EEMFlow+ (A8)
from oracle import eemflow_plus_oracle as P   # noqa: E402


def plus_sd(seed, cin):
    from eemflow_amd.eemflow_plus import EEMFlow_cdc
    from eemflow_amd.plus_weights import seeded_from_shapes
    shapes = {k: tuple(v.shape) for k, v in EEMFlow_cdc("", 3, cin).state_dict().items()}
    return O.to_torch_sd(seeded_from_shapes(shapes, seed))


def test_plus_warp_family(golden):
    g = golden("eemflow_plus_128x192.npz")
    x, flo = torch.from_numpy(g["w_x"]), torch.from_numpy(g["w_flo"])
    assert np.array_equal(P.warping_layer_no_div(x, flo).numpy(), g["w_no_div"])       # incl. the >= 1.0 mask: bit-exact
    assert np.array_equal(P.torch_warp(x, flo).numpy(), g["w_torch_warp"])
    assert np.array_equal(P.warp_align_true(x, flo).numpy(), g["w_align_true"])
    inp = torch.from_numpy(g["up_in"]).clone()
    up = P.upsample2d_flow_as(inp, (10, 12), if_rate=True)
    assert np.array_equal(up.numpy(), g["up_out"]) and np.array_equal(inp.numpy(), g["up_in_after"])   # in-place quirk


@pytest.mark.parametrize("tag", ["128x192", "100x150_c15"])
def test_plus_forward(golden, tag):
    g = golden(f"eemflow_plus_{tag}.npz")
    h, w = g["hw"].tolist()
    cin = int(g["cin"])
    sd = plus_sd(int(g["seed"]), cin)
    if "keys" in g.files:
        assert list(sd.keys()) == g["keys"].tolist() and len(sd) == 136
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w, bins=cin))
    # The reference's `grid_sample(ones) >= 1.0` warp mask flips on 1-ulp differences, and oneDNN's conv rounding
    # depends on the thread count: with the generator's 4 threads the oracle is BIT-IDENTICAL to the reference; with
    # another thread count the reference ITSELF moves by up to ~0.1 px on a third of the fine-level pixels.
    nt = torch.get_num_threads()
    torch.set_num_threads(4)
    try:
        with torch.no_grad():
            preds, st = P.eemflow_plus_forward(sd, e1, e2)
    finally:
        torch.set_num_threads(nt)
    assert st["pad"] == g["pad"].tolist()
    got = torch.stack(preds).numpy()
    np.testing.assert_allclose(got[:3], g["preds"][:3], atol=1e-5)          # levels 6, 5, 4: before the flips matter
    if os.cpu_count() and os.cpu_count() >= 4:
        assert np.array_equal(got, g["preds"])


@pytest.mark.parametrize("tag", ["128x192", "100x150_c15"])
def test_plus_levels_teacher_forced(golden, tag):
    """Each level of the oracle from the REFERENCE's own flow_init of that level (tests/golden/make_golden_plus_levels.py: hooks on
    the reference's cdc_model / decoders): with the discontinuous `>= 1.0` mask fed identical coordinates, every level holds the flow
    tolerance on its own, whatever the thread count - unlike the chained forward above."""
    g = golden(f"eemflow_plus_levels_{tag}.npz")
    h, w = g["hw"].tolist()
    cin = int(g["cin"])
    sd = plus_sd(int(g["seed"]), cin)
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(int(g["input_seed"]), int(g["batch"]), h, w, bins=cin))
    with torch.no_grad():
        _, st = P.eemflow_plus_forward(sd, e1, e2, keep=True)
        for l in (5, 4, 3, 2):
            fin = torch.from_numpy(g[f"flow_in{l}"]).clone()
            init = P.upsample2d_flow_as(fin, st["f1"][l].shape[-2:], if_rate=True)
            np.testing.assert_allclose(init.numpy(), g[f"flow_init{l}"], atol=1e-6)
            np.testing.assert_allclose(fin.numpy(), 2 * g[f"flow_in{l}"], rtol=1e-7)               # in-place doubling
            up, fl = P.level_from_init(sd, l, st["f1"][l], st["f2"][l], torch.from_numpy(g[f"flow_init{l}"]))
            np.testing.assert_allclose(up.numpy(), g[f"flow_up{l}"], atol=1e-4, err_msg=f"flow_up{l}")
            np.testing.assert_allclose(fl.numpy(), g[f"flow{l}"], atol=1e-4, err_msg=f"flow{l}")
