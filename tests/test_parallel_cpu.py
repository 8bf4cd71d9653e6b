"""world_size-2 gloo test of the one-process-per-GPU plumbing (frame sharding, barrier, MAX-over-ranks)."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from eemflow_amd import parallel

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, lr, w) == (rank, rank, world)
    lo, hi = parallel.shard_frames(11, r, w)
    parallel.barrier()
    seconds = 1.0 + rank            # rank 1 is the slow one
    rate, slowest = parallel.aggregate_throughput(hi - lo, seconds)
    import torch
    # data-parallel gradient exchange: per-rank gradients of equal shards -> mean over the global batch
    g = torch.full((1000,), float(rank + 1))
    parallel.average_gradients(g)
    assert torch.allclose(g, torch.full((1000,), 1.5))
    wts = torch.arange(10.0) * (rank + 1)
    parallel.broadcast_weights(wts, src=0)
    assert torch.equal(wts, torch.arange(10.0))
    q.put((rank, lo, hi, rate, slowest, parallel.max_over_ranks(rank * 10.0)))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def _dp_worker(rank, world, port, q):
    """Data-parallel step on real gradients: every rank differentiates the oracle's loss on ITS shard of the batch (torch autograd on
    the CPU restatement - the library has no CPU path), the flat gradients go through parallel.average_gradients, and every rank
    must end up with the gradient of the loss over the GLOBAL batch (train_mvsec.py:215 means over it)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.set_num_threads(2)
    from eemflow_amd.weights import seeded_state_dict, synthetic_gt, synthetic_voxel_pair
    from oracle import eemflow_oracle as O
    from oracle import train_oracle as T
    parallel.init_distributed("gloo")
    b, h, w = 4, 64, 64
    sd = O.to_torch_sd(seeded_state_dict(5))
    e1, e2 = (torch.from_numpy(a) for a in synthetic_voxel_pair(6, b, h, w))
    gt, valid = (torch.from_numpy(a) for a in synthetic_gt(7, b, h, w))
    lo, hi = parallel.shard_frames(b, rank, world)
    _, _, grads, _ = T.loss_and_grads(sd, e1[lo:hi], e2[lo:hi], gt[lo:hi], valid[lo:hi])
    flat = torch.cat([grads[k].reshape(-1) for k in sd])
    parallel.average_gradients(flat)
    _, _, full, _ = T.loss_and_grads(sd, e1, e2, gt, valid)
    ref = torch.cat([full[k].reshape(-1) for k in sd])
    q.put((rank, float((flat - ref).abs().max()), float(ref.abs().max()), flat.numel()))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_global_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, scale, n in out:
        assert n == 714352 and scale > 1e-3 and err < 1e-5 * scale + 1e-7, (rank, err, scale)


def test_per_rank_batch_split():
    from eemflow_amd.cli import per_rank_batch
    assert per_rank_batch(6, 1) == 6 and per_rank_batch(8, 8) == 1 and per_rank_batch(64, 8) == 8
    import pytest
    with pytest.raises(SystemExit):
        per_rank_batch(6, 4)                                    # nn.DataParallel would give ragged shards; the all-reduce mean needs equal ones


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, rate0, slow0, m0), (r1, lo1, hi1, rate1, slow1, m1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)              # disjoint, contiguous, covers all frames
    assert rate0 == rate1 == 11 / 2.0 and slow0 == slow1 == 2.0 and m0 == m1 == 10.0


def test_shard_frames_properties():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_frames(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    assert parallel.max_over_ranks(3.5) == 3.5
    assert parallel.aggregate_throughput(10, 2.0) == (5.0, 2.0)


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bench_never_measures_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` on a node with fewer GPUs (none here): refused before anything is launched; a launcher whose world
    size disagrees with --gpus is refused too; and when bench.py does start its own ranks (the sharing switch lets it, on a box with no
    GPU the ranks then fail), the children's failure is bench.py's exit code - no JSON line on any of these paths."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout
    import torch
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"], {"EEM_DIST_SHARE_GPU": "1"})
        assert r.returncode != 0 and "torch.distributed.run" in r.stderr and "{" not in r.stdout


def test_local_device_index_is_one_process_per_gpu(monkeypatch):
    import torch
    from eemflow_amd import parallel
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.delenv("EEM_DIST_SHARE_GPU", raising=False)
    assert parallel.local_device_index(1) == 1
    with pytest.raises(RuntimeError, match="one process per GPU"):
        parallel.local_device_index(2)
    monkeypatch.setenv("EEM_DIST_SHARE_GPU", "1")
    assert parallel.local_device_index(3) == 1


def test_numa_pinning_helper(tmp_path, monkeypatch):
    """parallel.pin_host_threads_to_gpu_numa: sysfs cpulist parsing, intersection with the process's own mask, never fatal."""
    import os
    from eemflow_amd import parallel
    assert parallel.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parallel.parse_cpulist("") == []
    assert parallel.pin_host_threads_to_gpu_numa(0, sysfs=str(tmp_path)) is None        # no GPU / no sysfs entry: nothing happens
    before = os.sched_getaffinity(0)
    monkeypatch.setenv("EEM_NO_NUMA_PIN", "1")
    assert parallel.pin_host_threads_to_gpu_numa(0) is None
    assert os.sched_getaffinity(0) == before
