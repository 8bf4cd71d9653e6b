"""One rank of a 2-process data-parallel run on a 1-GPU box (EEM_DIST_SHARE_GPU=1: both ranks on cuda:0, gloo all-reduce of the
device gradient) - started by tests/test_gpu_dp_processes.py through `python -m torch.distributed.run`.  Runs the PRODUCT's
data-parallel code with world > 1: EEMFlowTrainer.step (mode `trainer`) or `eemflow_amd.cli train` (mode `cli`), and leaves what
the parent compares in <out>/rank<r>.npz.  Not collected by pytest (no test_ prefix)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402


def flat_weights(model):
    return torch.cat([v.detach().reshape(-1).float().cpu() for v in model.state_dict().values()]).numpy()


def run_trainer(out, b, h, w, steps):
    from eemflow_amd import EEMFlow, parallel
    from eemflow_amd.train import EEMFlowTrainer
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    rank, local_rank, world = parallel.init_distributed()
    dev = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(133).items()})
    net = net.to(dev).train()
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(131, b, h, w))      # the GLOBAL batch; this rank takes its slice
    gt, va = (torch.from_numpy(a) for a in synthetic_gt(132, b, h, w))
    lo, hi = parallel.shard_frames(b, rank, world)
    sl = slice(lo, hi)
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    losses, grads = [], []
    for _ in range(steps):
        loss, _, _ = tr.step(e1[sl].to(dev), e2[sl].to(dev), gt[sl].to(dev), va[sl].to(dev))
        losses.append(loss)
        grads.append(tr.grad.detach().cpu().numpy().copy())                         # after the all-reduce: the averaged gradient
    tr.sync_parameters()
    np.savez(os.path.join(out, f"rank{rank}.npz"), losses=np.array(losses), weights=flat_weights(net), grads=np.stack(grads),
             world=world, backend=torch.distributed.get_backend())
    parallel.barrier(dev)
    torch.distributed.destroy_process_group()


def run_cli(out, root, config, engine_threads, model_name="EEMFlow", iters=2):
    from eemflow_amd import cli
    torch.manual_seed(11)                                            # rank 0's initial weights are broadcast anyway
    argv = ["train", "--data_root", root, "--save_root", out, "--lr", "1e-4", "--wd", "1e-5", "-bs", "2", "--train_iters", str(iters),
            "--val_iters", "1", "--config", config, "--model_name", model_name]
    if engine_threads:
        argv += ["-n", str(engine_threads)]
    cli.main(argv)
    run = cli.LAST_RUN
    if run["trainer"] is not None:
        run["trainer"].sync_parameters()
    params = torch.cat([p.detach().reshape(-1).float().cpu() for p in run["model"].parameters()]).numpy()
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
    np.savez(os.path.join(out, f"rank{run['rank']}.npz"), weights=flat_weights(run["model"]), params=params, world=run["world"],
             iteration=run["iteration"], engine=run["engine"], backend=backend)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def run_rccl1(out, b, h, w):
    """ONE rank under torch.distributed.run with EEM_DIST_BACKEND=nccl and EEM_DIST_FORCE=1: the data-parallel step's collectives on
    RCCL with itself as the only peer - library load, device-bound process group, all-reduce of the flat gradient inside
    EEMFlowTrainer.step, weight broadcast, barrier with device_ids, the MAX / SUM reductions of bench.py's reporting."""
    from eemflow_amd import EEMFlow, parallel
    from eemflow_amd.train import EEMFlowTrainer
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    rank, local_rank, world = parallel.init_distributed()
    assert torch.distributed.is_initialized() and world == 1 and parallel.exchange_active()
    dev = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(dev)
    net = EEMFlow("", 5, 5)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(133).items()})
    net = net.to(dev).train()
    net.change_imagesize((h, w))
    e1, e2 = (torch.from_numpy(a).to(dev) for a in synthetic_voxel_pair(131, b, h, w))
    gt, va = (torch.from_numpy(a).to(dev) for a in synthetic_gt(132, b, h, w))
    tr = EEMFlowTrainer(net, lr=1e-3, wdecay=5e-5, epsilon=1e-8, num_steps=20, clip=1.0)
    calls = {"n": 0}
    real = torch.distributed.all_reduce

    def counted(t, *a, **k):
        calls["n"] += 1
        return real(t, *a, **k)
    torch.distributed.all_reduce = counted
    loss, _, _ = tr.step(e1, e2, gt, va)                              # all-reduces tr.grad over RCCL (SUM over one rank: unchanged)
    torch.distributed.all_reduce = real
    grad = tr.grad.detach().clone()
    probe = torch.arange(714352, device=dev, dtype=torch.float32)    # the flat gradient's size
    parallel.average_gradients(probe)
    parallel.broadcast_weights(probe)
    parallel.barrier(dev)
    value, slowest = parallel.aggregate_throughput(7.0, 2.0, dev)
    torch.cuda.synchronize(dev)
    np.savez(os.path.join(out, "rank0.npz"), loss=loss, grad=grad.cpu().numpy(), probe_ok=bool(torch.equal(probe.cpu(), torch.arange(714352.0))),
             backend=torch.distributed.get_backend(), allreduce_calls=calls["n"], value=value, slowest=slowest,
             ipc_env=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "trainer":
        run_trainer(out, *(int(v) for v in sys.argv[3:7]))
    elif mode == "cli":
        run_cli(out, sys.argv[3], sys.argv[4], int(sys.argv[5]), *(sys.argv[6:7]), *(int(v) for v in sys.argv[7:8]))
    elif mode == "rccl1":
        run_rccl1(out, *(int(v) for v in sys.argv[3:6]))
    else:
        raise SystemExit(f"unknown mode {mode}")
