"""A16 with world > 1: the PRODUCT's data-parallel step in two real processes (train_EEMFlow_HREM.py:116-118 -> one process per GPU,
train_mvsec.py:215 -> mean over the global batch).

The GPU box has one MI355X and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and exchange the device gradient
over gloo (EEM_DIST_SHARE_GPU=1, eemflow_amd/parallel.py) - everything else is the code an 8-GPU job runs: torch.distributed.run,
parallel.init_distributed, EEMFlowTrainer.step's average_gradients on the gradient eemflow_forward_backward produced, `cli train`
with its DistributedSampler / per-rank batch / weight broadcast / rank-0 checkpoint, and bench.py's own launch of its ranks.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "dp_worker.py")
DEV = "cuda:0"


def child_env(share=True):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["OMP_NUM_THREADS"] = "4"
    if share:
        env["EEM_DIST_SHARE_GPU"] = "1"
    else:
        env.pop("EEM_DIST_SHARE_GPU", None)
    return env


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_two(args, timeout=600, nproc=2, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), WORKER, *[str(a) for a in args]]
    r = subprocess.run(cmd, env=env or child_env(), capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        # one more try on another port ONLY when the launcher itself lost the race for the port found free above (torch.distributed.run
        # binds it later; a rendezvous that fails this way fails before any of the code under test runs).  Anything else - a worker's
        # assertion, a crash, a numerical difference - fails here with the first attempt's output.
        first = r.stdout[-3000:] + r.stderr[-3000:]
        port_race = any(k in first for k in ("EADDRINUSE", "ddress already in use", "errno: 98", "RendezvousConnectionError",
                                             "DistNetworkError"))
        assert port_race, "worker failed (no rendezvous / bind error in its output, so no retry):\n" + first
        cmd[cmd.index("--master-port") + 1] = str(free_port())
        r = subprocess.run(cmd, env=env or child_env(), capture_output=True, text=True, timeout=timeout)
        assert r.returncode == 0, "first attempt (port race):\n" + first + "\nsecond attempt:\n" + r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_rccl_process_group_of_one_rank_runs_the_data_parallel_collectives(tmp_path):
    """backend "nccl" (= RCCL) has to work on this box before an 8-GPU node runs it: ONE rank under torch.distributed.run joins a
    device-bound RCCL process group (EEM_DIST_BACKEND=nccl, EEM_DIST_FORCE=1, HSA_ENABLE_IPC_MODE_LEGACY=0 as everywhere) and the
    trainer's step all-reduces its 2.86 MB flat gradient through it; broadcast, barrier(device_ids) and the reporting reductions too.
    The gradient equals the single-process one bit for bit (SUM over one rank)."""
    from eemflow_amd import EEMFlow
    from eemflow_amd.train import EEMFlowTrainer
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    b, h, w = 2, 260, 346
    env = child_env(share=False)
    env["EEM_DIST_BACKEND"] = "nccl"
    env["EEM_DIST_FORCE"] = "1"
    launch_two(["rccl1", tmp_path, b, h, w], nproc=1, env=env)
    r0 = np.load(os.path.join(tmp_path, "rank0.npz"))
    assert str(r0["backend"]) == "nccl" and int(r0["allreduce_calls"]) == 1 and bool(r0["probe_ok"])
    assert str(r0["ipc_env"]) == "0" and float(r0["value"]) == 3.5 and float(r0["slowest"]) == 2.0
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(133).items()})
    net = net.to(DEV).train()
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(131, b, h, w))
    gt, va = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(132, b, h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    loss, _, _ = tr.step(e1, e2, gt, va)
    g = tr.grad.cpu().numpy()
    assert abs(float(r0["loss"]) - loss) < 1e-6 * max(1.0, abs(loss))
    rel = np.linalg.norm(r0["grad"] - g) / np.linalg.norm(g)
    assert rel < 2e-5, rel                                    # (weight gradients: split-K atomics, order-dependent in the last bits)


@pytest.mark.parametrize("b,h,w", [(4, 260, 346), (2, 720, 1280)])
def test_trainer_step_two_processes_equals_single_process(tmp_path, b, h, w, monkeypatch):
    """EEMFlowTrainer.step under 2 ranks (half the batch each) against the same trainer on the whole batch in this process: the
    mean of the ranks' losses is the global loss, the all-reduced gradient is the global-batch gradient, the replicas end
    bit-identical and track the single-process weights."""
    from eemflow_amd import EEMFlow
    from eemflow_amd.train import EEMFlowTrainer
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    steps = 2
    monkeypatch.setenv("EEM_WINO4_LAYERS", "7")      # batch b here, b/2 per rank: pin the Winograd form (it follows the batch); ranks inherit
    launch_two(["trainer", tmp_path, b, h, w, steps])
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in (0, 1))
    assert int(r0["world"]) == 2 and str(r0["backend"]) == "gloo"
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(133).items()})
    net = net.to(DEV).train()
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(DEV) for a in synthetic_voxel_pair(131, b, h, w))
    gt, va = (torch.from_numpy(a).to(DEV) for a in synthetic_gt(132, b, h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    for i in range(steps):
        loss, _, _ = tr.step(e1, e2, gt, va)
        assert abs(0.5 * (r0["losses"][i] + r1["losses"][i]) - loss) < 2e-5 * max(1.0, abs(loss)), i
        g = tr.grad.cpu().numpy()
        assert np.array_equal(r0["grads"][i], r1["grads"][i]), i                     # every rank holds the same averaged gradient
        if i == 0:                                                                   # same weights on both sides only at step 0
            rel = np.linalg.norm(r0["grads"][0] - g) / np.linalg.norm(g)
            assert rel < 1e-4, rel
    tr.sync_parameters()
    wf = torch.cat([v.reshape(-1).float().cpu() for v in net.state_dict().values()]).numpy()
    assert np.array_equal(r0["weights"], r1["weights"])                              # replicas bit-identical after clip + AdamW
    d = np.abs(r0["weights"] - wf)
    assert float((d > 2e-5).mean()) < 2e-3 and float(d.max()) < 2.5e-3, float(d.max())


def _hrem_train_tree(root):
    from eemflow_amd import hrem
    for sub in ("dt1/000001", "dt1/000002"):
        d = os.path.join(root, "dataset/HREM/train", sub)
        os.makedirs(d)
        seed = 7 + int(sub[-1])
        hrem.write_events_npz(os.path.join(d, "events1.npz"), hrem.synthetic_hrem_events(seed, 20000, 720, 1280))
        hrem.write_events_npz(os.path.join(d, "events2.npz"), hrem.synthetic_hrem_events(seed + 1, 20000, 720, 1280))
        hrem.write_flo(os.path.join(d, "flow.flo"), hrem.synthetic_flow(seed + 2, 720, 1280))


@pytest.mark.parametrize("threads", [0, 2])
def test_cli_train_two_processes_equals_single_process(tmp_path, threads):
    """`cli train -bs 2` under torch.distributed.run with 2 ranks (per-rank batch 1, DistributedSampler, weight broadcast, rank-0
    checkpoint) against the same command in one process: two samples = the same global batch every step, so the two runs differ
    by summation order only.  No augmentation in the config (its flips draw from numpy's global generator)."""
    from eemflow_amd import cli
    root = str(tmp_path)
    _hrem_train_tree(root)
    cfg = json.loads(json.dumps(cli.DEFAULT_CONFIG))
    del cfg["data_loader"]["train"]["args"]["aug_params"]
    cfg_path = os.path.join(root, "cfg.json")
    json.dump(cfg, open(cfg_path, "w"))
    out2 = os.path.join(root, "two")
    os.makedirs(out2)
    r = launch_two(["cli", out2, root, cfg_path, threads])
    r0, r1 = (np.load(os.path.join(out2, f"rank{k}.npz")) for k in (0, 1))
    assert int(r0["world"]) == 2 and int(r0["iteration"]) == 2 and int(r1["iteration"]) == 2
    assert np.array_equal(r0["weights"], r1["weights"])                              # replicas identical after two steps
    run2 = os.path.join(out2, "exp_HREM_meshflow/EEMFlow_dt1/lr0.000100_we0.000010")
    assert sorted(os.listdir(run2)) == ["config.json", "lasted_ckpt.pth.tar", "train.log"]   # one writer
    log = open(os.path.join(run2, "train.log")).read().strip().splitlines()
    assert len(log) == 2, log                                                        # rank 0's lines only, one per epoch
    ck2 = torch.load(os.path.join(run2, "lasted_ckpt.pth.tar"), weights_only=False)
    out1 = os.path.join(root, "one")
    os.makedirs(out1)
    torch.manual_seed(11)
    run1 = cli.main(["train", "--data_root", root, "--save_root", out1, "--lr", "1e-4", "--wd", "1e-5", "-bs", "2", "--train_iters", "2",
                     "--val_iters", "1", "--config", cfg_path])
    ck1 = torch.load(os.path.join(run1, "lasted_ckpt.pth.tar"), weights_only=False)
    assert ck1["epoch"] == ck2["epoch"] == 1 and ck1["iteration"] == ck2["iteration"] == 2
    diff = torch.cat([(ck1["state_dict"][k] - ck2["state_dict"][k]).abs().reshape(-1) for k in ck1["state_dict"]])
    assert float(diff.max()) < 3e-4 and float(diff.median()) < 1e-6, (float(diff.max()), float(diff.median()))
    # the checkpoint holds what rank 0's replica holds
    w_ck = torch.cat([v.reshape(-1).float() for v in ck2["state_dict"].values()]).numpy()
    assert np.array_equal(w_ck, r0["weights"])


def test_cli_train_eraft_two_processes_equals_the_mean_of_per_sample_gradients(tmp_path):
    """`cli train --model_name eraft -bs 2` (train_EEMFlow_HREM.py:30-32,116-118) under two ranks: the autograd engine's
    data-parallel branch (harness.TrainRaftEvents._train_iters_autograd: flat all-reduce of the still-scaled gradients) against
    what it must equal - ONE process that runs each of the two samples as a batch of one (a replica's BatchNorm sees only its own
    sample, as under nn.DataParallel), averages the two gradients, clips and takes torch's AdamW step."""
    from eemflow_amd import cli, harness
    from eemflow_amd.hrem import HREMEventFlow
    from eemflow_amd.train import sequence_loss
    root = str(tmp_path)
    _hrem_train_tree(root)
    cfg = json.loads(json.dumps(cli.DEFAULT_CONFIG))
    del cfg["data_loader"]["train"]["args"]["aug_params"]
    cfg_path = os.path.join(root, "cfg.json")
    json.dump(cfg, open(cfg_path, "w"))
    out2 = os.path.join(root, "two")
    os.makedirs(out2)
    launch_two(["cli", out2, root, cfg_path, 0, "eraft", 1], timeout=900)
    r0, r1 = (np.load(os.path.join(out2, f"rank{k}.npz")) for k in (0, 1))
    assert int(r0["world"]) == 2 and str(r0["engine"]) == "autograd" and str(r0["backend"]) == "gloo"
    assert int(r0["iteration"]) == 1 and int(r1["iteration"]) == 1
    assert np.array_equal(r0["params"], r1["params"])                # replicas' PARAMETERS identical (BatchNorm buffers are per replica)
    run2 = os.path.join(out2, "exp_HREM_meshflow/eraft_dt1/lr0.000100_we0.000010")
    assert sorted(os.listdir(run2)) == ["config.json", "lasted_ckpt.pth.tar", "train.log"]
    ck2 = torch.load(os.path.join(run2, "lasted_ckpt.pth.tar"), weights_only=False)
    assert ck2["iteration"] == 1
    # the single-process statement of the same step
    torch.manual_seed(11)
    model = cli.build_model("eraft", cfg, training=True).to(DEV)
    tcfg = dict(cfg["data_loader"]["train"]["args"])
    tcfg.update({'type': 'train', 'event_interval': 'dt1', 'batch_size': 2})
    ds = HREMEventFlow(args=tcfg, train=True, root=root, device=torch.device(DEV))
    s0 = ds[0]
    model.change_imagesize(tuple(int(v) for v in s0['event_volume_old'].shape[-2:]))
    model.train()
    p0 = torch.cat([p.detach().reshape(-1).float().cpu() for p in model.parameters()]).numpy()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 1e-4, 1 + 100, pct_start=0.05, cycle_momentum=False, anneal_strategy='linear')
    opt.zero_grad()
    for i in range(2):
        smp = ds[i]
        e1, e2 = smp['event_volume_old'][None].to(DEV).float(), smp['event_volume_new'][None].to(DEV).float()
        _, preds = model(e1, e2)
        gt, va = harness._target_like(preds[-1], smp['flow'][None].to(DEV).float(), smp['valid'][None].to(DEV).float())
        loss, _ = sequence_loss(preds, gt, va, 0.8)
        (0.5 * loss).backward()                                      # gradients accumulate: the mean over the two replicas
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    g_ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1).float().cpu()
                       for p in model.parameters()]).numpy()
    opt.step()
    sched.step()
    p1 = torch.cat([p.detach().reshape(-1).float().cpu() for p in model.parameters()]).numpy()
    step_ref, step_dp = p1 - p0, r0["params"] - p0
    assert float(np.abs(step_ref).max()) > 1e-6                      # the step moved the weights
    # AdamW's first step is lr * g / (|g| + eps): sign-like.  An element whose gradient is round-off of an exact zero (a conv bias in front
    # of an instance norm, a unit no sample activates) takes a step of either sign in ANY two runs - the weight gradients' atomics add
    # in another order every time.  Those elements are named by the REFERENCE gradient (|g| below 1e-4 of its rms: round-off of the sums
    # that formed it) and left out; on every other element the original bounds hold (the bounds are not widened).
    rms = float(np.sqrt(np.mean(g_ref.astype(np.float64) ** 2)))
    sure = np.abs(g_ref) > 1e-4 * rms
    bad = np.abs(step_dp - step_ref) > 0.05 * np.abs(step_ref).max()
    print(f"eraft dp step: {1 - sure.mean():.4f} of the elements at round-off level, bad among the others {bad[sure].mean():.2e} "
          f"(among all {bad.mean():.2e}), relative step difference {np.linalg.norm((step_dp - step_ref)[sure]) / np.linalg.norm(step_ref[sure]):.3e}")
    assert float(sure.mean()) > 0.5, float(sure.mean())          # (14 % of E-RAFT's 5.3 M elements sit at round-off level: exact zeros included)
    assert float(bad[sure].mean()) < 5e-3, float(bad[sure].mean())
    assert np.linalg.norm((step_dp - step_ref)[sure]) / np.linalg.norm(step_ref[sure]) < 0.05


def _bench(args, share, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=child_env(share), capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself and rank 0's line says
    n_gpus 2; the training mode reports the all-reduce it timed."""
    r, line = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--preheat", "10", "--cpu-seconds", "0", "--no-other-rows",
                      "--no-side-rows", "--height", "260", "--width", "346"], share=True)
    assert r.returncode == 0 and line is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0 and line["scaling"] == "weak"
    one, l1 = _bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--preheat", "10", "--cpu-seconds", "0", "--no-other-rows",
                      "--no-side-rows", "--height", "260", "--width", "346"], share=False)
    assert one.returncode == 0 and l1["n_gpus"] == 1
    r, line = _bench(["--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1", "--height", "260", "--width", "346",
                      "--batch", "4"], share=True)
    assert r.returncode == 0 and line is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["allreduce_us"] > 0 and line["allreduce_bytes"] == 714352 * 4
    assert np.isfinite(line["final_loss"]) and line["config"]["parallelism"].startswith("dp2")


def test_bench_eraft_workload_two_ranks():
    """BASELINE configs[4] has an N-rank product path: `bench.py --workload eraft --gpus 2` (inference: replicas; --mode train: the
    autograd step with one all-reduce of the flat parameter gradient) launches its own ranks and reports n_gpus 2; without the sharing
    switch it refuses to measure two ranks on one GPU."""
    small = ["--workload", "eraft", "--height", "128", "--width", "160", "--batch", "2", "--iters", "3", "--steps", "3", "--warmup", "1"]
    r, line = _bench(["--gpus", "2", *small], share=True)
    assert r.returncode == 0 and line is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["unit"] == "frames/s" and line["value"] > 0 and "configs[4]" in line["metric"]
    assert line["config"]["batch_per_gpu"] == 2 and line["config"]["parallelism"].startswith("replicas x2")
    r, line = _bench(["--gpus", "2", "--mode", "train", *small], share=True)
    assert r.returncode == 0 and line is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["unit"] == "samples/s" and line["allreduce_us"] > 0 and line["allreduce_bytes"] > 4 * 5e6
    assert np.isfinite(line["final_loss"]) and line["config"]["parallelism"].startswith("dp2") and line["config"]["backend"] == "gloo"
    if torch.cuda.device_count() < 2:
        r, line = _bench(["--gpus", "2", *small], share=False, timeout=300)
        assert r.returncode != 0 and line is None and "refusing" in r.stderr


def test_bench_refuses_more_ranks_than_gpus():
    """Without the sharing switch, asking for more GPUs than the node has is an error - never a silent 1-GPU number."""
    n = torch.cuda.device_count()
    r, line = _bench(["--gpus", str(n + 1), "--steps", "2", "--warmup", "1"], share=False, timeout=300)
    assert r.returncode != 0 and line is None and "refusing" in r.stderr
