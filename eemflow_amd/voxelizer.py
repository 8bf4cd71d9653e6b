"""Event voxelization on MI355X behind the reference's loader interface.

`EventSequence` and `EventSequenceToVoxelGrid_Pytorch` keep the constructor arguments and call
convention of loader/loader_utils.py:352-397 and :429-537, so HREM/MVSEC dataset code
(loader/HREM.py:140-144,226-232) can use them unchanged.  The host side only orders, scales and
shifts the timestamps (float64 numpy, as the reference does); the voting and the normalisation
run in libeemflow_hip.so.  There is no CPU voting path: without a GPU this raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class EventSequence(object):
    """(N,4) float64 events [t, x, y, p]; sorted by time on construction (loader_utils.py:352-397)."""

    def __init__(self, dataframe, params, features=None, timestamp_multiplier=None, convert_to_relative=False):
        if dataframe is not None and hasattr(dataframe, "to_numpy"):
            self.feature_names = dataframe.columns.values
            self.features = dataframe.to_numpy()
        else:
            self.feature_names = np.array(['ts', 'x', 'y', 'p'], dtype=object)
            self.features = np.zeros([1, 4]) if features is None else features
        self.image_height = params['height']
        self.image_width = params['width']
        if not self.is_sorted():
            self.sort_by_timestamp()
        if timestamp_multiplier is not None:
            self.features[:, 0] *= timestamp_multiplier          # in place, like the reference
        if convert_to_relative:
            self.absolute_time_to_relative()

    def get_sequence_only(self):
        return self.features

    def __len__(self):
        return len(self.features)

    def __add__(self, sequence):
        return EventSequence(dataframe=None, features=np.concatenate([self.features, sequence.features]),
                             params={'height': self.image_height, 'width': self.image_width})

    def is_sorted(self):
        t = self.features[:, 0]
        return bool(np.all(t[:-1] <= t[1:]))

    def sort_by_timestamp(self):
        if len(self.features[:, 0]) > 0:
            self.features = self.features[np.argsort(self.features[:, 0])]

    def absolute_time_to_relative(self):
        start_ts = self.features[:, 0].min()
        assert start_ts == self.features[0, 0]
        self.features[:, 0] -= start_ts


DEFERRED = "deferred"


def _norm_code(normalize):
    """normalize: False (raw grid), True (the reference's normalised grid) or "deferred" - the grid stays raw and the four floats behind it
    hold {mean, sd, scale, any} for a consumer that normalises as it reads (EEMFlow.forward_many(..., deferred_norm=True))."""
    if isinstance(normalize, str):
        if normalize != DEFERRED:
            raise ValueError(f"normalize: True, False or '{DEFERRED}'; got {normalize!r}")
        return 2
    return 1 if normalize else 0


def _grid_buffers(k, num_bins, height, width, dev, deferred):
    """k (num_bins,H,W) fp32 tensors; with `deferred` each is a view of a flat buffer that is four floats longer (the record)."""
    n = num_bins * height * width
    if not deferred:
        return [torch.empty(num_bins, height, width, dtype=torch.float32, device=dev) for _ in range(k)]
    return [torch.empty(n + 4, dtype=torch.float32, device=dev)[:n].view(num_bins, height, width) for _ in range(k)]


def has_norm_record(t):
    """Is this a deferred grid: a contiguous tensor that starts its storage and whose storage is EXACTLY its elements plus the four-float
    normalisation record (what _grid_buffers allocates)?  "Room for four more floats" alone would also accept frame i of a stacked
    [N,C,H,W] tensor or any over-allocated buffer, whose "record" is the next frame's first voxels; the exact shape of the
    allocation is the tag - views that keep the storage ([None], .float() of a float tensor, .view) keep it."""
    return (t.is_contiguous() and t.storage_offset() == 0
            and t.untyped_storage().nbytes() == (t.numel() + 4) * t.element_size())


def norm_record(t):
    """The record {mean, sd, scale, any} behind a deferred grid (a 4-element view of its storage)."""
    if not has_norm_record(t):
        raise ValueError("norm_record: the tensor has no room for a record behind it")
    return torch.as_strided(t, (4,), (1,), t.storage_offset() + t.numel())


def voxelize_device(events, num_bins, height, width, normalize=True, out=None, return_indices=False):
    """Voxelize events that already live on the GPU: `events` is an (N,4) float64 CUDA tensor [t, x, y, p], time-sorted, timestamps
    already scaled / made relative (what EventSequence holds).  No host copy, no synchronisation; `out` (num_bins,H,W) fp32 is reused
    when given.  The device-resident form of EventSequenceToVoxelGrid_Pytorch.__call__ (loader_utils.py:447-537)."""
    if not (isinstance(events, torch.Tensor) and events.is_cuda and events.dtype == torch.float64 and events.dim() == 2
            and events.shape[1] == 4 and events.is_contiguous()):
        raise _lib.EEMFlowHipError("voxelize_device: events must be a contiguous (N,4) float64 CUDA tensor")
    dev = events.device
    n = events.shape[0]
    with torch.no_grad(), torch.cuda.device(dev):
        code = _norm_code(normalize)
        grid = out if out is not None else _grid_buffers(1, num_bins, height, width, dev, code == 2)[0]
        if tuple(grid.shape) != (num_bins, height, width) or grid.dtype != torch.float32 or not grid.is_contiguous():
            raise ValueError("voxelize_device: out must be a contiguous (num_bins,H,W) fp32 tensor")
        if code == 2 and not has_norm_record(grid):
            raise ValueError("voxelize_device: a deferred grid needs four floats of storage behind it")
        il = ir = None
        if return_indices:
            il = torch.empty(n, dtype=torch.int64, device=dev)
            ir = torch.empty(n, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().eemflow_voxelize(
            events.data_ptr(), n, num_bins, height, width, code, grid.data_ptr(),
            il.data_ptr() if return_indices else None, ir.data_ptr() if return_indices else None, _lib.current_stream_ptr(dev)))
    return (grid, il, ir) if return_indices else grid


def voxelize_pair_device(events1, events2, num_bins, height, width, normalize=True, out=None):
    """Both event sets of a sample in one launch sequence (eemflow_voxelize_pair): (N1,4) and (N2,4) float64 CUDA tensors ->
    a (2,num_bins,H,W) fp32 tensor, [0] = events1's grid.  The same values as two voxelize_device calls."""
    for ev in (events1, events2):
        if not (isinstance(ev, torch.Tensor) and ev.is_cuda and ev.dtype == torch.float64 and ev.dim() == 2 and ev.shape[1] == 4
                and ev.is_contiguous()):
            raise _lib.EEMFlowHipError("voxelize_pair_device: events must be contiguous (N,4) float64 CUDA tensors")
    dev = events1.device
    code = _norm_code(normalize)
    with torch.no_grad(), torch.cuda.device(dev):
        if code == 2:                                            # two separate buffers, each with its record behind it
            grids = list(out) if out is not None else _grid_buffers(2, num_bins, height, width, dev, True)
            if len(grids) != 2 or any(tuple(g.shape) != (num_bins, height, width) or g.dtype != torch.float32 or not has_norm_record(g) for g in grids):
                raise ValueError("voxelize_pair_device: deferred grids are two (num_bins,H,W) fp32 tensors with four floats of storage behind each")
        else:
            grids = out if out is not None else torch.empty(2, num_bins, height, width, dtype=torch.float32, device=dev)
            if tuple(grids.shape) != (2, num_bins, height, width) or grids.dtype != torch.float32 or not grids.is_contiguous():
                raise ValueError("voxelize_pair_device: out must be a contiguous (2,num_bins,H,W) fp32 tensor")
        _lib.check(_lib.lib().eemflow_voxelize_pair(events1.data_ptr(), events1.shape[0], events2.data_ptr(), events2.shape[0], num_bins,
                                                    height, width, code, grids[0].data_ptr(), grids[1].data_ptr(),
                                                    _lib.current_stream_ptr(dev)))
    return grids


MAX_SETS_PER_CALL = 32


def voxelize_many_device(event_sets, num_bins, height, width, normalize=True, out=None):
    """Up to 32 event sets of one sensor size in ONE launch sequence (eemflow_voxelize_many) - e.g. both volumes of every sample of an
    `EEMFlow.forward_many` call: a list of (N_k,4) float64 CUDA tensors -> a list of (num_bins,H,W) fp32 tensors (`out`: the tensors to
    fill), each the same values as `voxelize_device` gives for that set alone."""
    event_sets = list(event_sets)
    if not 1 <= len(event_sets) <= MAX_SETS_PER_CALL:
        raise ValueError(f"voxelize_many_device: 1..{MAX_SETS_PER_CALL} event sets per call, got {len(event_sets)}")
    for ev in event_sets:
        if not (isinstance(ev, torch.Tensor) and ev.is_cuda and ev.dtype == torch.float64 and ev.dim() == 2 and ev.shape[1] == 4
                and ev.is_contiguous()):
            raise _lib.EEMFlowHipError("voxelize_many_device: events must be contiguous (N,4) float64 CUDA tensors")
    dev = event_sets[0].device
    k = len(event_sets)
    code = _norm_code(normalize)
    with torch.no_grad(), torch.cuda.device(dev):
        grids = list(out) if out is not None else _grid_buffers(k, num_bins, height, width, dev, code == 2)
        if len(grids) != k or any(tuple(g.shape) != (num_bins, height, width) or g.dtype != torch.float32 or not g.is_contiguous() for g in grids):
            raise ValueError("voxelize_many_device: out must be one contiguous (num_bins,H,W) fp32 tensor per event set")
        if code == 2 and not all(has_norm_record(g) for g in grids):
            raise ValueError("voxelize_many_device: deferred grids need four floats of storage behind each")
        ptr = ctypes.c_void_p * k
        _lib.check(_lib.lib().eemflow_voxelize_many(k, ptr(*[e.data_ptr() for e in event_sets]), (ctypes.c_int64 * k)(*[e.shape[0] for e in event_sets]),
                                                    num_bins, height, width, code, ptr(*[g.data_ptr() for g in grids]),
                                                    _lib.current_stream_ptr(dev)))
    return grids


class EventSequenceToVoxelGrid_Pytorch(object):
    def __init__(self, num_bins, gpu=False, gpu_nr=0, normalize=True, forkserver=True):
        if forkserver:
            try:
                torch.multiprocessing.set_start_method('forkserver')
            except RuntimeError:
                pass
        self.num_bins = num_bins
        self.normalize = normalize
        self.return_on_gpu = bool(gpu)          # the reference returns a CPU tensor when gpu=False
        self.device = torch.device('cuda:' + str(gpu_nr))

    def __call__(self, event_sequence, return_indices=False):
        width, height = event_sequence.image_width, event_sequence.image_height
        if isinstance(event_sequence.features, torch.Tensor) and event_sequence.features.is_cuda:
            # events already on the GPU (a loader that keeps them resident): no astype, no host-to-device copy
            res = voxelize_device(event_sequence.features.to(self.device, torch.float64).contiguous(), self.num_bins, height, width,
                                  self.normalize, return_indices=return_indices)
            if not self.return_on_gpu:
                res = (res[0].cpu(),) + tuple(res[1:]) if return_indices else res.cpu()
            return res
        events = np.ascontiguousarray(event_sequence.features.astype('float'))
        assert events.shape[1] == 4
        assert self.num_bins > 0 and width > 0 and height > 0
        if not torch.cuda.is_available():
            raise _lib.EEMFlowHipError("EventSequenceToVoxelGrid_Pytorch: no GPU - the voxelizer has no CPU path here")
        n = events.shape[0]
        code = _norm_code(self.normalize)
        if code == 2 and not self.return_on_gpu:
            raise _lib.EEMFlowHipError("normalize='deferred' grids live on the GPU (gpu=True): their record is read by the first convolution")
        with torch.no_grad(), torch.cuda.device(self.device):
            ev = torch.from_numpy(events).to(self.device)
            grid = _grid_buffers(1, self.num_bins, height, width, self.device, code == 2)[0]       # deferred: raw grid + its record
            il = ir = None
            if return_indices:
                il = torch.empty(n, dtype=torch.int64, device=self.device)
                ir = torch.empty(n, dtype=torch.int64, device=self.device)
            _lib.check(_lib.lib().eemflow_voxelize(
                ev.data_ptr(), n, self.num_bins, height, width, code, grid.data_ptr(),
                il.data_ptr() if return_indices else None, ir.data_ptr() if return_indices else None,
                _lib.current_stream_ptr(self.device)))
        if not self.return_on_gpu:
            grid = grid.cpu()
        if return_indices:
            return grid, il, ir
        return grid
    # (normalize="deferred" - raw grids with the normalisation record behind them, for EEMFlow.forward_many(..., deferred_norm=True) -
    # needs gpu=True in every form of the call: the grids stay on the GPU, where the record's consumer is)

    def many(self, sequences):
        """`[self(s) for s in sequences]` - up to 32 sequences of one sensor size - by one launch sequence."""
        sequences = list(sequences)
        width, height = sequences[0].image_width, sequences[0].image_height
        if any((s.image_width, s.image_height) != (width, height) for s in sequences):
            raise ValueError("many: the event sequences must share the sensor size")
        if not torch.cuda.is_available():
            raise _lib.EEMFlowHipError("EventSequenceToVoxelGrid_Pytorch: no GPU - the voxelizer has no CPU path here")
        with torch.no_grad(), torch.cuda.device(self.device):
            evs = []
            for seq in sequences:
                f = seq.features
                if isinstance(f, torch.Tensor):
                    evs.append(f.to(self.device, torch.float64).contiguous())
                else:
                    evs.append(torch.from_numpy(np.ascontiguousarray(f.astype('float'))).to(self.device))
            grids = voxelize_many_device(evs, self.num_bins, height, width, self.normalize)
        if self.normalize == DEFERRED and not self.return_on_gpu:
            raise _lib.EEMFlowHipError("normalize='deferred' grids live on the GPU (gpu=True): their record is read by the first convolution")
        return [g if self.return_on_gpu else g.cpu() for g in grids]

    def pair(self, sequence_old, sequence_new):
        """The two volumes of a sample - `self(sequence_old), self(sequence_new)` - by one launch sequence instead of two."""
        width, height = sequence_old.image_width, sequence_old.image_height
        if (sequence_new.image_width, sequence_new.image_height) != (width, height):
            raise ValueError("pair: the two event sequences must share the sensor size")
        if not torch.cuda.is_available():
            raise _lib.EEMFlowHipError("EventSequenceToVoxelGrid_Pytorch: no GPU - the voxelizer has no CPU path here")
        with torch.no_grad(), torch.cuda.device(self.device):
            evs = []
            for seq in (sequence_old, sequence_new):
                f = seq.features
                if isinstance(f, torch.Tensor):
                    evs.append(f.to(self.device, torch.float64).contiguous())
                else:
                    evs.append(torch.from_numpy(np.ascontiguousarray(f.astype('float'))).to(self.device))
            grids = voxelize_pair_device(evs[0], evs[1], self.num_bins, height, width, self.normalize)
        if self.normalize == DEFERRED:
            if not self.return_on_gpu:
                raise _lib.EEMFlowHipError("normalize='deferred' grids live on the GPU (gpu=True): their record is read by the first convolution")
            return grids[0], grids[1]
        if not self.return_on_gpu:
            grids = grids.cpu()
        return grids[0], grids[1]
