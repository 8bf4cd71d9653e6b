"""Slim evaluation / training harness around the GPU path.

Keeps what the reference's harness does ON the path and its log lines, drops cv2 / pandas / imageio / git:

  load_checkpoint / save_checkpoint   test_EEMFlow_HREM.py:59-66, train_EEMFlow_HREM.py:127-130:
                                      torch.save({'epoch', 'state_dict'}) '.pth.tar', keys optionally 'module.'-prefixed
  TestRaftEvents.test_multi_sequence  test_mvsec.py:538-671: per sequence, every `stride`-th sample: change_imagesize,
                                      model(events1, events2), flow_error on the last prediction, the reference's summary
                                      lines; returns the mean AEE over the sequences
  TrainRaftEvents.train_iters         train_mvsec.py:229-286: model.change_imagesize, one optimisation step per batch.
                                      engine="fused" (default): EEMFlowTrainer - loss, backward, [RCCL all-reduce], clip +
                                      AdamW + OneCycle inside the library; engine="autograd": the reference's literal sequence
                                      (fetch_optimizer's torch AdamW + OneCycleLR, GradScaler, model(im1, im2) -> sequence_loss
                                      -> scaler.scale(loss).backward() -> clip_grad_norm_ -> scaler.step) through the model's
                                      torch.autograd.Function - for callers that bring their own optimizer or loss

Samples come from a dataset with the reference's dict keys ('event_volume_old', 'event_volume_new', 'flow', 'valid',
'event_valid'); tensors are moved to the model's device.  One process per GPU (eemflow_amd.parallel), not
nn.DataParallel.  Unlike the reference, save_checkpoint can also store the step count of the schedule.
"""
import collections
import itertools
import sys

import torch

from .metrics import flow_error, flow_error_from_sums, flow_error_sums, flow_error_sums_many
from . import parallel
from .train import EEMFlowTrainer, sequence_loss


class Logger:
    """write_line(text, also_print) as the reference's logger is used; lines are kept in `.lines`."""

    def __init__(self, path=None, verbose=True):
        self.path, self.lines, self.verbose = path, [], verbose

    def write_line(self, text, verbose=True):
        self.lines.append(text)
        if self.path:
            with open(self.path, "a") as f:
                f.write(text + "\n")
        if verbose and self.verbose:
            print(text)
            sys.stdout.flush()


def load_checkpoint(path, model, map_location="cpu", with_iteration=False):
    """Load {'epoch', 'state_dict'}; 'module.' prefixes of nn.DataParallel checkpoints are stripped. Returns the epoch, or with
    with_iteration=True (epoch, iteration): the schedule position save_checkpoint stores (0 for the reference's own checkpoints)."""
    states = torch.load(path, map_location=map_location, weights_only=False)
    sd = collections.OrderedDict((k.replace('module.', ''), v) for k, v in states['state_dict'].items())
    model.load_state_dict(sd)
    if with_iteration:
        return states.get('epoch', 0), int(states.get('iteration', 0))
    return states.get('epoch', 0)


def save_checkpoint(path, model, epoch, trainer=None, module_prefix=False, iteration=None):
    """Write the reference's checkpoint layout.  With a trainer, the device-resident weights are synced into the
    module first and the schedule position is stored under 'iteration' (the reference does not keep it); `iteration` stores the
    position of a loop that has no fused trainer (the autograd engine)."""
    if trainer is not None:
        trainer.sync_parameters()
    sd = model.state_dict()
    if module_prefix:
        sd = collections.OrderedDict(('module.' + k, v) for k, v in sd.items())
    out = {'epoch': epoch, 'state_dict': collections.OrderedDict((k, v.detach().cpu()) for k, v in sd.items())}
    if trainer is not None:
        out['iteration'] = trainer.iteration
    elif iteration is not None:
        out['iteration'] = int(iteration)
    torch.save(out, path)


def _device_of(model):
    return next(model.parameters()).device


class TestRaftEvents:
    """Evaluation loop of test_mvsec.py:538-671 on a dataset object (HREMEventFlow-like: change_test_sequence, __len__,
    __getitem__ -> sample dict)."""
    __test__ = False                      # not a pytest class

    def __init__(self, dataset, image_size, logger=None, is_car=False):
        self.dataset = dataset
        self.image_size = image_size
        self.logger = logger or Logger()
        self.is_car = is_car

    def run_network(self, model, sample, dev):
        e1 = sample['event_volume_old'].to(dev)[None].float()
        e2 = sample['event_volume_new'].to(dev)[None].float()
        _, preds = model(events1=e1, events2=e2)
        return preds[-1]

    def test_multi_sequence(self, model, epoch=0, sequence_list=(), stride=10, frames_in_flight=1, loader_threads=0, coalesce=1):
        """The evaluation loop of test_mvsec.py:580-597.  It reads the LAST prediction of every sample only (run_network, :1455): a
        model that can skip forming the earlier ones (ERAFT.final_only) does so for the duration of the call."""
        had = getattr(model, "final_only", None)
        if had is not None:
            model.final_only = True
        try:
            return self._test_multi_sequence(model, epoch, sequence_list, stride, frames_in_flight, loader_threads, coalesce)
        finally:
            if had is not None:
                model.final_only = had

    def _test_multi_sequence(self, model, epoch=0, sequence_list=(), stride=10, frames_in_flight=1, loader_threads=0, coalesce=1):
        """coalesce > 1 (a model with forward_many - EEMFlow, EEMFlow_cdc, ERAFT - and a dataset with get_samples): that many samples are read, voxelized by
        ONE voxelizer launch sequence and handed to ONE model.forward_many call - n independent batch-1 samples riding a batch-n chain of
        launches, every sample in its own tensors; raw volumes with a normalisation record (HREMEventFlow(deferred_norm=True)) are
        normalised by the first convolution as it reads them.  Chunks alternate over min(frames_in_flight, 2) replicas / streams.  Same
        per-sample lines in the same order; the numbers agree with the one-sample loop to fp32 round-off (the batched chain uses the
        F(4x4) Winograd form on every stride-1 layer).
        frames_in_flight > 1: that many replicas of the model (model.replicate()) take the samples round robin, each on a HIP
        stream of its own - loading and voxelizing sample i + 1 and the forwards of the samples before it overlap; a sample's
        statistics are fetched (and its line printed, in order) when frames_in_flight - 1 newer samples have been enqueued.
        Same numbers as the sequential loop: every sample runs the same kernels on the same data.
        loader_threads > 0: that many host threads read and voxelize samples ahead (dataset[idx] on the thread's current stream, an
        event hands the sample over to the consuming stream) - the file reads and npz decompression release the GIL, and at 1280x720
        they are 20 ms of a sample's 20.3 ms."""
        model.change_imagesize(self.image_size)
        model.eval()
        dev = _device_of(model)
        self.logger.write_line("test in stride {:d}".format(stride), True)
        sparse = getattr(self.dataset, "evaluation_type", "dense") == "sparse"
        mean_aee, mean_out, aee_list, out_list = 0., 0., [], []
        nfl = max(1, int(frames_in_flight))
        co = max(1, int(coalesce))
        if co > 1:
            if not (hasattr(model, "forward_many") and hasattr(self.dataset, "get_samples")):
                raise ValueError("coalesce > 1 needs a model with forward_many (EEMFlow, EEMFlow_cdc, ERAFT) and a dataset with get_samples (HREMEventFlow)")
            co = min(co, 16)
            nfl = min(nfl, 2)                                    # two batched chains fill the chip; more only share it
        replicas, streams = [model], [torch.cuda.current_stream(dev)]
        hint_before = getattr(model, "frames_in_flight", 1)
        if nfl > 1:
            model.frames_in_flight = nfl
            replicas += [model.replicate(nfl) for _ in range(nfl - 1)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
        pool = None
        if loader_threads > 0:                                   # ONE pool for the whole evaluation (threads are not re-created per sequence)
            import concurrent.futures
            pool = concurrent.futures.ThreadPoolExecutor(max_workers=loader_threads)
        with torch.no_grad():
            for sequence in sequence_list:
                acc = collections.defaultdict(float)
                iters, n_points = 0, 0
                self.dataset.change_test_sequence(sequence)
                pending = collections.deque()

                def retire():
                    nonlocal iters, n_points
                    idx, k, sums, keep = pending.popleft()
                    with torch.cuda.stream(streams[k]):
                        aee, p1, p3, n_points, s_ee, aee_gt, s_gt = flow_error_from_sums(sums)
                    for name, v in (("aee", aee), ("sum", s_ee), ("aee_gt", aee_gt), ("sum_gt", s_gt), ("p1", p1), ("p3", p3)):
                        acc[name] += v
                    iters += 1
                    print('{:05d} / {:05d}  AEE: {:2.6f}  meanAEE:{:2.6f} 3 - mean %AEE: {:.6f}'.format(
                        idx + 1, len(self.dataset), aee, acc["aee"] / iters, 1. - acc["p3"] / iters))

                indices = [idx for idx in range(len(self.dataset)) if idx % stride == 0]
                futures = collections.deque()
                if pool is not None and co == 1:                 # (the coalesced loop reads its chunks through dataset.get_samples: no
                    def load(idx):                               # per-sample futures are submitted that nobody would consume - ADVICE round 5)
                        with torch.cuda.device(dev):
                            sample = self.dataset[idx]
                            ready = torch.cuda.Event()
                            ready.record(torch.cuda.current_stream(dev))
                        return sample, ready
                    ahead = iter(indices)
                    for idx in itertools.islice(ahead, 2 * loader_threads):
                        futures.append(pool.submit(load, idx))
                count = 0
                for c0 in (range(0, len(indices), co) if co > 1 else ()):
                    chunk = indices[c0:c0 + co]
                    k = count % nfl
                    count += 1
                    with torch.cuda.stream(streams[k]):
                        samples = self.dataset.get_samples(chunk)
                        frames = [(s_['event_volume_old'].to(dev)[None].float(), s_['event_volume_new'].to(dev)[None].float()) for s_ in samples]
                        deferred = all(s_.get('deferred_norm', False) for s_ in samples)
                        if deferred:                             # (.float() / [None] keep the storage: the record stays behind the volume)
                            outs = replicas[k].forward_many(frames, deferred_norm=True)
                        else:
                            outs = replicas[k].forward_many(frames)
                        f_ests = [preds[-1] for _, preds in outs]
                        f_gts = [s_['flow'].to(dev)[None].float() for s_ in samples]
                        evs_ = [s_['event_valid'].to(dev).sum(0) for s_ in samples] if (sparse and all('event_valid' in s_ for s_ in samples)) else None
                        all_sums = flow_error_sums_many(f_gts, f_ests, evs_, is_car=self.is_car,        # the chunk's statistics by one launch
                                                        evaluation_type="sparse" if evs_ is not None else "dense")
                        for i_, (idx, sample) in enumerate(zip(chunk, samples)):
                            pending.append((idx, k, all_sums[i_], (sample, f_ests[i_], f_gts[i_], evs_[i_] if evs_ is not None else None)))
                    while len(pending) >= nfl * co:
                        retire()
                for idx in (indices if co == 1 else ()):
                    k = count % nfl
                    count += 1
                    with torch.cuda.stream(streams[k]):
                        if pool is not None:
                            sample, ready = futures.popleft().result()
                            streams[k].wait_event(ready)
                            for nidx in itertools.islice(ahead, 1):
                                futures.append(pool.submit(load, nidx))
                        else:
                            sample = self.dataset[idx]
                        f_est = self.run_network(replicas[k], sample, dev)
                        f_gt = sample['flow'].to(dev)[None].float()
                        ev = sample['event_valid'].to(dev).sum(0) if ('event_valid' in sample and sparse) else None
                        sums = flow_error_sums(f_gt, f_est, ev, is_car=self.is_car, evaluation_type="sparse" if ev is not None else "dense")
                    pending.append((idx, k, sums, (sample, f_est, f_gt, ev)))     # the tensors stay alive until the sample retires
                    while len(pending) >= nfl:
                        retire()
                while pending:
                    retire()
                iters = max(iters, 1)
                self.logger.write_line("-------------------test_sequence_{:s}------------------".format(sequence), True)
                self.logger.write_line(
                    "Mean AEE: {:.6f}, sum AEE: {:.6f}, Mean AEE_gt: {:.6f}, sum AEE_gt: {:.6f}, 1 - mean %AEE: {:.6f}, "
                    "3 - mean %AEE: {:.6f}, # pts: {:.6f}".format(acc["aee"] / iters, acc["sum"] / iters, acc["aee_gt"] / iters,
                                                                  acc["sum_gt"] / iters, 1. - acc["p1"] / iters,
                                                                  1. - acc["p3"] / iters, n_points), True)
                mean_aee += acc["aee"] / iters
                mean_out += 1. - acc["p3"] / iters
                aee_list.append(acc["aee"] / iters)
                out_list.append(1. - acc["p3"] / iters)
        if pool is not None:
            pool.shutdown()
        model.frames_in_flight = hint_before
        self.logger.write_line("-------------------------------------------------------", True)
        self.logger.write_line("-----------------Test after {:d} epoch-----------------".format(epoch), True)
        for name, a, o in zip(sequence_list, aee_list, out_list):
            self.logger.write_line("{:s}: Mean AEE: {:.6f},  3 - mean %AEE: {:.6f}".format(name, a, o), True)
        self.logger.write_line("-------------------------------------------------------", True)
        n = max(len(sequence_list), 1)
        self.logger.write_line("Average points: Mean AEE: {:.6f},  3 - mean %AEE: {:.6f}".format(mean_aee / n, mean_out / n), True)
        return mean_aee / n


def _target_like(pred, flow_gt, valid):
    """HREM's training target is the 16x16 mesh flow (HREM.py:254-255).  A model that predicts at another size (E-RAFT and EEMFlow+
    at full resolution) is trained against that mesh flow taken to its size the way the dataset's own evaluation branch does it
    (HREM.py:264-267: bilinear, align_corners=False; valid = finite and non-zero) - the library's resize kernel, not a torch op."""
    if tuple(flow_gt.shape[-2:]) == tuple(pred.shape[-2:]):
        return flow_gt, valid
    from . import _lib
    b, _, h, w = flow_gt.shape
    oh, ow = int(pred.shape[-2]), int(pred.shape[-1])
    src = flow_gt.contiguous()
    full = torch.empty(b, 2, oh, ow, device=src.device, dtype=torch.float32)
    _lib.check(_lib.lib().eemflow_upsample_bilinear(src.data_ptr(), full.data_ptr(), 2 * b, h, w, oh, ow, _lib.current_stream_ptr(src.device)))
    ok = (~torch.isinf(full[:, 0])) & (~torch.isinf(full[:, 1])) & (torch.linalg.norm(full, dim=1) > 0)
    return full, ok.float()


class TrainRaftEvents:
    """Training loop of train_mvsec.py:229-286 (one process per GPU; batches are this rank's shard)."""

    def __init__(self, loader, image_size, lr=1e-4, wdecay=5e-5, epsilon=1e-8, num_steps=1000000, clip=1.0, gamma=0.8,
                 logger=None, print_freq=100, engine="fused", mixed_precision=True, start_iteration=0):
        if engine not in ("fused", "autograd"):
            raise ValueError("engine must be 'fused' or 'autograd'")
        self.loader, self.image_size = loader, image_size
        self.opt = dict(lr=lr, wdecay=wdecay, epsilon=epsilon, num_steps=num_steps, clip=clip, gamma=gamma)
        self.logger = logger or Logger()
        self.print_freq = print_freq
        self.engine, self.mixed_precision = engine, mixed_precision
        self.trainer = None
        self.optimizer = self.scheduler = self.scaler = None
        self.iteration = int(start_iteration)                   # resume: the OneCycle schedule continues where the checkpoint stopped

    def fetch_optimizer(self, model):
        """train_mvsec.py:178-183."""
        o = self.opt
        self.optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=o["lr"], weight_decay=o["wdecay"],
                                           eps=o["epsilon"])
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(self.optimizer, o["lr"], o["num_steps"] + 100, pct_start=0.05,
                                                             cycle_momentum=False, anneal_strategy='linear')

    def _train_iters_autograd(self, model, start_epoch, val_iters):
        """The body of train_mvsec.py:241-258, statement for statement; in data-parallel jobs the parameter gradients are
        averaged by one RCCL all-reduce of a flat buffer before the clip (what DataParallel's gather/scatter amounts to)."""
        dev = _device_of(model)
        if self.optimizer is None:
            self.fetch_optimizer(model)
            self.scaler = torch.amp.GradScaler("cuda", enabled=self.mixed_precision)
            if self.iteration:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")             # "scheduler.step() before optimizer.step()": positions only
                    for _ in range(self.iteration):
                        self.scheduler.step()
        done = 0
        for batch in self.loader:
            self.optimizer.zero_grad()
            e1 = batch['event_volume_old'].to(dev).float()
            e2 = batch['event_volume_new'].to(dev).float()
            _, flow_list = model(e1, e2)
            flow_gt, valid = _target_like(flow_list[-1], batch['flow'].to(dev).float(), batch['valid'].to(dev).float())
            loss, metrics = sequence_loss(flow_list, flow_gt, valid, self.opt["gamma"])
            self.scaler.scale(loss).backward()
            if parallel.exchange_active():
                # the still-SCALED gradients are exchanged, then unscaled: an overflow on one rank reaches every rank through the
                # sum, so all ranks record the same found_inf, skip the same step and keep the same scale
                params = [p for p in model.parameters() if p.grad is not None]
                flat = torch.cat([p.grad.reshape(-1) for p in params])
                parallel.average_gradients(flat)
                off = 0
                for p in params:
                    p.grad.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
            self.scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), self.opt["clip"])
            lr = self.optimizer.param_groups[0]["lr"]
            self.scaler.step(self.optimizer)
            self.scheduler.step()
            self.scaler.update()
            done += 1
            self.iteration += 1
            if done % self.print_freq == 0 or done == 1:
                self.logger.write_line("[epoch {:d}, {:6d}] loss {:.6f} epe {:.4f} lr {:.3e}".format(
                    start_epoch, self.iteration, loss.item(), metrics["epe"], lr), True)
            if val_iters is not None and done >= val_iters:
                break
        return model

    def train_iters(self, model, start_epoch=0, val_iters=None):
        if self.image_size is None:                              # padder sized from the data (cli: un-cropped HREM frames)
            ds = getattr(self.loader, "dataset", None)
            first = ds[0] if ds is not None else next(iter(self.loader))
            self.image_size = tuple(int(v) for v in first['event_volume_old'].shape[-2:])
        model.change_imagesize(self.image_size)
        model.train()
        dev = _device_of(model)
        if self.engine == "autograd":
            return self._train_iters_autograd(model, start_epoch, val_iters)
        if self.trainer is None:
            self.trainer = EEMFlowTrainer(model, **self.opt)
            self.trainer.iteration = self.iteration
        done = 0
        for batch in self.loader:
            e1 = batch['event_volume_old'].to(dev).float()
            e2 = batch['event_volume_new'].to(dev).float()
            loss, metrics, _ = self.trainer.step(e1, e2, batch['flow'].to(dev).float(), batch['valid'].to(dev).float())
            done += 1
            if done % self.print_freq == 0 or done == 1:
                self.logger.write_line("[epoch {:d}, {:6d}] loss {:.6f} epe {:.4f} lr {:.3e}".format(
                    start_epoch, self.trainer.iteration, loss, metrics["epe"], metrics["lr"]), True)
            if val_iters is not None and done >= val_iters:
                break
        return model
