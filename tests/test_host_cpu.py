"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host mirrors of the
reference interfaces behave like the reference (against golden vectors), nothing falls back to CPU."""
import os
import re

import numpy as np
import pytest
import torch

from eemflow_amd import EEMFlow, EventSequence, InputPadder, _lib
from eemflow_amd.weights import eemflow_param_shapes, seeded_state_dict, strip_module_prefix
from oracle import eemflow_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "eemflow_hip.h")).read()
    declared = set(re.findall(r"\b((?:eemflow|eraft|eemplus|eemop)_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    handle = _lib.lib()                      # resolves every symbol or raises
    assert handle.eemflow_abi_version() == 1
    assert handle.eemflow_last_error() is not None


def test_kernels_with_asm_loads_use_no_scratch(tmp_path):
    """gconv16's and conv_wnc's weight fragments arrive through asm loads whose completion the compiler does not track: an instance that
    spills would have a register parked in scratch and reused while its load is in flight (how gconv16's stride-2 form first died on the
    GPU).  Every instance in the built code objects must have a private segment of 0 bytes."""
    import subprocess
    from eemflow_amd.build import CSRC, build_library
    llvm = "/opt/rocm/lib/llvm/bin"
    for stem, kernel, at_least in (("gconv16", "gconv16_kernel", 60), ("conv_wnc", "wnc_kernel", 4)):
        obj = os.path.join(CSRC, "build", stem + ".o")
        if not os.path.exists(obj):
            build_library(verbose=False)
        fat, co = str(tmp_path / f"{stem}.fat.bin"), str(tmp_path / f"{stem}.co")
        subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, str(tmp_path / f"{stem}.host.o")], check=True)
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        kernels = re.findall(r"\.name:\s+(\S*" + kernel + r"\S*)", notes)
        sizes = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
        assert len(kernels) >= at_least and len(sizes) >= len(kernels), (stem, len(kernels))
        assert all(v == 0 for v in sizes), (stem, sorted(set(sizes)))


def test_header_cites_reference_interfaces():
    header = open(os.path.join(REPO, "include", "eemflow_hip.h")).read()
    assert header.count("Replaces:") >= 8 and "EEMFlow.py:122-183" in header and "loader_utils.py:447-537" in header


def test_state_dict_layout(golden):
    g = golden("state_dict_layout.npz")
    net = EEMFlow("", groups=5, n_first_channels=5)
    sd = net.state_dict()
    assert list(sd.keys()) == g["keys"].tolist()
    assert [list(v.shape) for v in sd.values()] == [s[s >= 0].tolist() for s in g["shapes"]]
    assert sum(v.numel() for v in sd.values()) == 714352 == int(g["nparams"])
    assert list(eemflow_param_shapes().keys()) == list(sd.keys())
    # DataParallel-prefixed checkpoints load after stripping, as test_EEMFlow_HREM.py:62-66 does
    pref = {"module." + k: torch.from_numpy(v) for k, v in seeded_state_dict(0).items()}
    net.load_state_dict(strip_module_prefix(pref))
    flat = net._flat_weights()
    assert flat.numel() == 714352 and torch.equal(flat[:720], torch.from_numpy(seeded_state_dict(0)["pconv1_1.0.weight"]).flatten())


def test_init_matches_reference_scheme():
    torch.manual_seed(0)
    net = EEMFlow("", groups=5, n_first_channels=5)
    assert all(float(p.abs().max()) == 0 for n, p in net.named_parameters() if n.endswith("bias"))
    w = net.decoder_1.conv1[0].weight
    assert abs(float(w.std()) - (2.0 / (69 * 9)) ** 0.5) < 2e-3           # kaiming_normal_, fan_in


def test_padder_matches_reference(golden):
    g = golden("pad.npz")
    for h, w, rate, mode, *pad in g["table"].tolist():
        assert InputPadder((h, w), mode="chairs" if mode == 0 else "sintel", eval_pad_rate=rate)._pad == pad
    p = InputPadder((5, 7), mode="chairs", eval_pad_rate=4)
    xp = p.pad(torch.from_numpy(g["x"]))[0]
    assert np.array_equal(xp.numpy(), g["x_padded"]) and np.array_equal(p.unpad(xp).numpy(), g["x_unpadded"])


def test_event_sequence_host_logic(golden):
    g = golden("voxel.npz")
    for name in ("n400_unsorted", "n20k", "n2_dt0"):
        ev = g[f"{name}_events"]
        seq = EventSequence(None, {"height": 48, "width": 64}, features=ev.copy(), timestamp_multiplier=1e6,
                            convert_to_relative=True)
        assert np.array_equal(seq.features, O.event_sequence(ev, 1e6, True))
        assert seq.is_sorted() and len(seq) == len(ev) and seq.features[0, 0] == 0
    a = EventSequence(None, {"height": 4, "width": 4}, features=np.array([[1.0, 0, 0, 1], [2.0, 1, 1, -1]]))
    assert len(a + a) == 4 and (a + a).is_sorted()
    assert EventSequence(None, {"height": 4, "width": 4}).features.shape == (1, 4)


def test_no_cpu_fallback():
    net = EEMFlow("", groups=5, n_first_channels=5).eval()
    net.change_imagesize((64, 64))
    with pytest.raises(_lib.EEMFlowHipError, match="no CPU path"):
        net(torch.zeros(1, 5, 64, 64), torch.zeros(1, 5, 64, 64))
    with pytest.raises(_lib.EEMFlowHipError):
        net.upsample_flow(torch.zeros(1, 2, 4, 4), (8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "eemflow_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libeemflow_hip.so")
    with pytest.raises(_lib.EEMFlowHipError, match="no CPU fallback"):
        _lib.lib()


def test_eraft_state_dict_layout(golden):
    from eemflow_amd.eraft import ERAFT
    g = golden("eraft_layout.npz")
    sd = ERAFT("", 5).state_dict()
    assert list(sd.keys()) == g["keys"].tolist() and len(sd) == 179
    assert [list(v.shape) for v in sd.values()] == [s[s >= 0].tolist() for s in g["shapes"]]
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == int(g["nparams"])


def test_onecycle_schedule_matches_torch():
    from eemflow_amd.train import OneCycleLinear
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-4, 300 + 100, pct_start=0.05, cycle_momentum=False,
                                                anneal_strategy='linear')           # train_mvsec.py:182-183
    mine = OneCycleLinear(1e-4, 300 + 100)
    for step in range(400):
        assert abs(opt.param_groups[0]["lr"] - mine.lr(step)) < 1e-12 * 1e4, step
        opt.step()
        if step < 399:
            sched.step()


def test_checkpoint_layout_round_trip(tmp_path):
    """{'epoch', 'state_dict'} '.pth.tar' files, 'module.' prefixes accepted (test_EEMFlow_HREM.py:59-66)."""
    import os
    import torch
    from eemflow_amd import EEMFlow
    from eemflow_amd.harness import load_checkpoint, save_checkpoint
    a, b = EEMFlow("", 5, 5), EEMFlow("", 5, 5)
    p = os.path.join(tmp_path, "ckpt.pth.tar")
    save_checkpoint(p, a, epoch=7, module_prefix=True)
    raw = torch.load(p, weights_only=False)
    assert raw["epoch"] == 7 and all(k.startswith("module.") for k in raw["state_dict"]) and len(raw["state_dict"]) == 66
    assert load_checkpoint(p, b) == 7
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert load_checkpoint(p, b, with_iteration=True) == (7, 0)          # the reference's layout: no schedule position

    class Position:                                                       # what save_checkpoint needs of a trainer
        iteration = 1234

        def sync_parameters(self):
            pass
    save_checkpoint(p, a, epoch=8, trainer=Position())
    assert load_checkpoint(p, b, with_iteration=True) == (8, 1234)


def test_resumed_trainer_continues_the_one_cycle_schedule():
    """TrainRaftEvents(start_iteration=n): the autograd engine's OneCycleLR stands where n steps would have left it, the fused engine's
    schedule is indexed by trainer.iteration (train_mvsec.py:178-183; the reference restarts the warm-up on resume)."""
    import torch
    from eemflow_amd.harness import TrainRaftEvents
    from eemflow_amd.train import OneCycleLinear as OneCycle
    n, lr, steps = 300, 4e-4, 1000
    lin = torch.nn.Linear(2, 2)
    ref_opt = torch.optim.AdamW(lin.parameters(), lr=lr)
    ref = torch.optim.lr_scheduler.OneCycleLR(ref_opt, lr, steps + 100, pct_start=0.05, cycle_momentum=False, anneal_strategy='linear')
    for _ in range(n):
        ref_opt.step()
        ref.step()
    tr = TrainRaftEvents(loader=[], image_size=(8, 8), lr=lr, num_steps=steps, engine="autograd", start_iteration=n)
    tr.fetch_optimizer(lin)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(tr.iteration):
            tr.scheduler.step()
    assert abs(tr.optimizer.param_groups[0]["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12
    assert abs(OneCycle(lr, steps + 100).lr(n) - ref_opt.param_groups[0]["lr"]) < 1e-9


def test_threaded_batch_loader_follows_the_dataloader_protocol():
    """eemflow_amd.loader.ThreadedBatchLoader (what `--num_workers` maps to): the batches of a sequential DataLoader, in order, with
    drop_last, with a sampler, over several epochs; seeded shuffles differ between epochs and cover every sample."""
    import time
    import torch
    from eemflow_amd.loader import ThreadedBatchLoader

    class Toy(torch.utils.data.Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            time.sleep(0.002 * ((i * 7) % 3))                      # uneven load times: order must not depend on them
            return {"x": torch.full((2, 3), float(i)), "i": torch.tensor(i), "name": "s%d" % i}
    ds = Toy()
    ref = list(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, drop_last=True))
    for epoch in range(2):
        got = list(ThreadedBatchLoader(ds, 4, threads=3))
        assert len(got) == len(ref) == 2 == len(ThreadedBatchLoader(ds, 4, threads=3))
        for a, b in zip(got, ref):
            assert torch.equal(a["x"], b["x"]) and torch.equal(a["i"], b["i"]) and a["name"] == list(b["name"])
    assert len(list(ThreadedBatchLoader(ds, 4, threads=2, drop_last=False))) == 3
    sampler = torch.utils.data.SequentialSampler(range(6))
    assert [b["i"].tolist() for b in ThreadedBatchLoader(ds, 3, sampler=sampler, threads=2)] == [[0, 1, 2], [3, 4, 5]]
    sh = ThreadedBatchLoader(ds, 5, shuffle=True, threads=2, drop_last=False)
    e1 = [i for b in sh for i in b["i"].tolist()]
    e2 = [i for b in sh for i in b["i"].tolist()]
    assert sorted(e1) == sorted(e2) == list(range(11)) and e1 != e2


def test_cli_flags_match_reference_scripts():
    """Flags and defaults of train_EEMFlow_HREM.py:139-156 / test_EEMFlow_HREM.py:124-140."""
    from eemflow_amd import cli
    p = cli.build_parser()
    a = p.parse_args(["train"])
    assert (a.train_iters, a.val_iters, a.lr, a.wd, a.batch_size, a.model_name, a.input_type, a.start_epoch, a.test_sequence) == \
        (6000000, 10000, 1e-5, 0, 6, "EEMFlow", "dt1", False, "indoor_flying2")
    b = p.parse_args(["test", "-bs", "4", "-int", "dt4", "-se", "-model", "eraft", "-sq", "s1", "-n", "2"])
    assert (b.train_iters, b.val_iters, b.lr, b.wd, b.batch_size, b.input_type, b.start_epoch, b.model_name, b.test_sequence, b.num_workers) == \
        (1000000, 3000, 1e-4, 1e-5, 4, "dt4", True, "eraft", "s1", 2)
    assert cli.DEFAULT_CONFIG["train"]["gamma"] == 0.8 and cli.DEFAULT_CONFIG["val_img_size"] == [720, 1280]
