"""world_size-2 gloo test of the one-process-per-GPU plumbing (frame sharding, barrier, MAX-over-ranks)."""
import os
import socket

import torch.multiprocessing as mp

from eemflow_amd import parallel


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
