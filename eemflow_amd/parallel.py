"""One-process-per-GPU plumbing for the EEMFlow hot path.

Inference shards along the batch/frame dimension with no data-path exchange (the reference's only
strategy is nn.DataParallel's batch scatter, train_EEMFlow_HREM.py:116-118): every rank owns a
replica and a disjoint slice of the frames; the only collectives are a barrier and a MAX-reduction
of the elapsed time for reporting.  Backend "nccl" is RCCL on ROCm; tests use "gloo" on CPU.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def share_gpu():
    """EEM_DIST_SHARE_GPU=1: several ranks may sit on one GPU (rank r -> device r mod #devices) and the process group is gloo, which
    all-reduces device tensors through the host.  RCCL refuses two ranks on one device, so this is how the data-parallel step of the
    product (EEMFlowTrainer.step, `cli train`, `bench.py --mode train`) is exercised with world > 1 on a 1-GPU box
    (tests/test_gpu_dp_processes.py); never set in production."""
    return os.environ.get("EEM_DIST_SHARE_GPU", "0") not in ("", "0")


def local_device_index(local_rank=None):
    """The GPU of this rank: LOCAL_RANK, one process per GPU.  A rank without a GPU of its own is an error (no silent sharing)
    unless EEM_DIST_SHARE_GPU is set."""
    if local_rank is None:
        local_rank = env_world()[1]
    n = torch.cuda.device_count()
    if n == 0:
        raise RuntimeError("eemflow_amd.parallel: no GPU visible to this rank")
    if local_rank >= n:
        if not share_gpu():
            raise RuntimeError(f"eemflow_amd.parallel: LOCAL_RANK {local_rank} but only {n} GPU(s) visible - one process per GPU")
        return local_rank % n
    return local_rank


def force_group():
    """EEM_DIST_FORCE=1: join a process group even as a job of ONE rank (launched by torch.distributed.run --nproc-per-node=1), so
    that a 1-GPU box runs the very collectives an 8-GPU job runs - RCCL's library load, the device-bound process group, the
    all-reduce / broadcast / barrier of the data-parallel step - with itself as the only peer (tests/test_gpu_dp_processes.py)."""
    return os.environ.get("EEM_DIST_FORCE", "0") not in ("", "0")


def init_distributed(backend=None):
    """Join the process group when launched with WORLD_SIZE > 1 (or forced, see force_group); returns (rank, local_rank, world).
    Backend: EEM_DIST_BACKEND, else "nccl" (= RCCL) with GPUs ("gloo" when ranks share a GPU, see share_gpu), "gloo" on CPU."""
    rank, local_rank, world = env_world()
    if (world > 1 or (force_group() and "RANK" in os.environ)) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("EEM_DIST_BACKEND") or None
        if backend is None:
            backend = "gloo" if (share_gpu() or not torch.cuda.is_available()) else "nccl"
        kwargs = {}
        if backend == "nccl":
            idx = local_device_index(local_rank)
            torch.cuda.set_device(idx)
            kwargs["device_id"] = torch.device("cuda", idx)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs local_cpulist) -> sorted list of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def pin_host_threads_to_gpu_numa(device_index, sysfs="/sys/bus/pci/devices"):
    """One process per GPU: keep this rank's host threads - every thread the process has now, and the ones created later (the loader's
    workers inherit the mask) - on the CPUs local to ITS GPU's PCIe
    root - on an 8-GPU node the ranks otherwise share one socket's memory controllers for their pinned staging buffers.  Reads the GPU's
    PCI address from torch and `local_cpulist` from sysfs; intersects with the mask the process already has (a cgroup's); returns the
    CPUs it pinned to, or None when anything is missing (no sysfs entry, an empty intersection) - never fatal.  EEM_NO_NUMA_PIN=1 skips it."""
    if os.environ.get("EEM_NO_NUMA_PIN", "0") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        p = torch.cuda.get_device_properties(device_index)
        addr = "{:04x}:{:02x}:{:02x}.0".format(p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        local = parse_cpulist(open(os.path.join(sysfs, addr, "local_cpulist")).read())
        allowed = sorted(set(local) & set(os.sched_getaffinity(0)))
        if not allowed:
            return None
        # every thread this process already has (sched_setaffinity(0, ...) moves the CALLING thread only: RCCL's proxy threads, torch's
        # intra-op pool and the HIP runtime's helpers exist by the time a rank knows its GPU) - threads created later inherit the mask
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), allowed)
            except OSError:                                  # a thread that exited between the listing and the call
                pass
        os.sched_setaffinity(0, allowed)
        return allowed
    except Exception:                                        # noqa: BLE001 - a missing sysfs entry must not stop a run
        return None


def exchange_active():
    """True when this process takes part in data-path collectives: a process group exists and has more than one rank (or the
    one-rank group was forced, see force_group)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_group())


def shard_frames(n_frames, rank, world):
    """Contiguous, disjoint, near-equal slice [start, stop) of n_frames for this rank."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (identity without a process group)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(units_this_rank, seconds_this_rank, device=None):
    """Whole-job rate: units processed by ALL ranks / the slowest rank's time."""
    total = sum_over_ranks(units_this_rank, device)
    slowest = max_over_ranks(seconds_this_rank, device)
    return total / slowest, slowest


def average_gradients(flat_grad):
    """Data-parallel gradient exchange: ONE all-reduce (RCCL on GPUs) of the flat fp32 gradient buffer, then
    divide by the world size - the mean over the global batch when every rank holds an equal shard, which is
    what the reference's single-process nn.DataParallel computes (train_mvsec.py:215 means over the whole
    scattered batch).  No-op without a process group."""
    if exchange_active():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        if dist.get_world_size() > 1:
            flat_grad.div_(dist.get_world_size())
    return flat_grad


def broadcast_weights(flat_weights, src=0):
    """Replicas start from rank `src`'s weights (replaces DataParallel's per-step re-broadcast by one at start)."""
    if exchange_active():
        dist.broadcast(flat_weights, src=src)
    return flat_weights
