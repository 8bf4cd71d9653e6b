"""HREM dataset front-end feeding the GPU voxelizer.

Mirrors the reference's `loader/HREM.py` + the file readers of `loader/loader_utils.py`:

  get_compressed_events(path)   loader_utils.py:26-42   events{1,2}.npz (x, y, t [ns], p in {0,1}) -> (N,4) float64
  read_flo(path)                loader_utils.py:54-65   Middlebury .flo -> (H,W,2) float32
  motion_propagate(flow, H, W)  HREM.py:41-101          16x16 mesh-flow ground truth (12-sample vertex medians, 5x5 median)
  HREMEventFlow(args, train)    HREM.py:128-274         same args dict, directory layout, sample dict keys

Host code is Python/numpy as in the reference (file parsing and a 16x16 gather are not GPU work); the event
volumes are voxelized by libeemflow_hip.so (EventSequenceToVoxelGrid_Pytorch of this package) and stay on the
GPU unless `to_cpu=True`.  `motion_propagate` is vectorised (the reference loops over Python dicts) and returns
the same values bit for bit (tests/test_data_rows.py).  The train split's augmentation is eemflow_amd.augmentor.FlowAugmentor
(the reference's no-resize path: random flips), built from args['aug_params'] as in HREM.py:146-150, or `augmentor=`.
"""
import os

import numpy as np
import torch

from . import _lib
from .voxelizer import EventSequence, EventSequenceToVoxelGrid_Pytorch

FLO_MAGIC = 202021.25


# ------------------------------------------------------------------------------------------------ files
def get_compressed_events(event_path):
    """events npz -> (N,4) float64 [t in seconds, x, y, p in {-1,+1}] (loader_utils.py:26-37)."""
    d = np.load(event_path)
    out = np.empty((d["t"].shape[0], 4), dtype=np.float64)
    out[:, 0] = d["t"] * 1e-9
    out[:, 1] = d["x"]
    out[:, 2] = d["y"]
    out[:, 3] = 2 * d["p"] - 1
    return out


def read_flo(flow_path):
    """Middlebury .flo -> (H,W,2) float32; None when the magic number is wrong (loader_utils.py:54-65)."""
    with open(flow_path, "rb") as f:
        head = np.fromfile(f, np.float32, count=1)
        if head.size != 1 or head[0] != np.float32(FLO_MAGIC):
            print('Magic number incorrect. Invalid .flo file')
            return None
        w, h = np.fromfile(f, np.int32, count=2)
        data = np.fromfile(f, np.float32, count=2 * int(w) * int(h))
    return np.resize(data, (int(h), int(w), 2))


def write_flo(flow_path, flow):
    """(H,W,2) float32 -> .flo (inverse of read_flo; the reference only reads)."""
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    with open(flow_path, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([flow.shape[1], flow.shape[0]], np.int32).tofile(f)
        flow.tofile(f)


def write_events_npz(path, events):
    """(N,4) [t seconds, x, y, p in {-1,+1}] -> the npz layout get_compressed_events reads (t in ns, p in {0,1}).
    p is stored signed: the reader computes 2*p - 1 in the array's own dtype (loader_utils.py:34), which wraps to 255
    for an unsigned p = 0 - in the reference too."""
    np.savez(path, x=events[:, 1].astype(np.uint16), y=events[:, 2].astype(np.uint16),
             t=np.round(events[:, 0] * 1e9).astype(np.int64), p=((events[:, 3] + 1) // 2).astype(np.int8))


def synthetic_flow(seed, h, w):
    """Smooth random flow field (H,W,2) float32 for tests / fixtures: a coarse random grid, bilinearly interpolated
    (only IEEE add / multiply, so the values do not depend on the libm / SIMD dispatch of the host) plus noise."""
    rng = np.random.default_rng(seed)
    gh, gw = 5, 7
    coarse = rng.normal(0.0, 4.0, (gh, gw, 2))
    y = np.arange(h, dtype=np.float64) * ((gh - 1) / max(h - 1, 1))
    x = np.arange(w, dtype=np.float64) * ((gw - 1) / max(w - 1, 1))
    y0 = np.minimum(y.astype(np.int64), gh - 2); x0 = np.minimum(x.astype(np.int64), gw - 2)
    fy = (y - y0)[:, None, None]; fx = (x - x0)[None, :, None]
    c00 = coarse[y0][:, x0]; c01 = coarse[y0][:, x0 + 1]; c10 = coarse[y0 + 1][:, x0]; c11 = coarse[y0 + 1][:, x0 + 1]
    flow = (c00 * (1 - fx) + c01 * fx) * (1 - fy) + (c10 * (1 - fx) + c11 * fx) * fy
    return (flow + rng.normal(0, 0.05, (h, w, 2))).astype(np.float32)


def flow_error_inputs(seed, h, w):
    """(gt (2,H,W), pred (2,H,W), event_img (1,H,W)) float32 with zero-flow and inf ground-truth patches - test inputs
    for the flow_error metrics."""
    rng = np.random.default_rng(seed + 1000)
    gt = synthetic_flow(seed, h, w).transpose(2, 0, 1).copy()
    gt[:, 10:20, 30:50] = 0.0
    gt[0, 40:44, 5:9] = np.inf
    pred = gt + rng.normal(0, 1.5, gt.shape).astype(np.float32)
    pred[np.isinf(pred)] = 0.0
    ev_img = ((rng.random((1, h, w)) < 0.3) * rng.integers(1, 5, (1, h, w))).astype(np.float32)
    return gt, pred, ev_img


def synthetic_hrem_events(seed, n, h, w, t_span=0.05):
    """(N,4) float64 events [t seconds (ns resolution), x, y, p in {-1,+1}], unsorted like a raw recording slice."""
    rng = np.random.default_rng(seed)
    t = np.round(rng.uniform(0, t_span, n) * 1e9) * 1e-9
    return np.stack([t, rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 2, n) * 2 - 1], axis=1).astype(np.float64)


# ------------------------------------------------------------------------------------------------ mesh flow
def motion_propagate(fflow, height, width, mesh_size=16, radius=3):
    """Mesh-flow ground truth of HREM.py:41-101, vectorised: every mesh vertex (i, j) takes the upper median of the
    flow at 4 mirrored offsets x `radius` radii around (mesh_rows*i, mesh_cols*j) (indices clamped to the image),
    then a 5x5 median over the replicate-padded 16x16 mesh.  Returns (x_mesh, y_mesh), float64 (16,16)."""
    fflow = np.asarray(fflow)
    u, v = fflow[..., 0], fflow[..., 1]
    mesh_cols, mesh_rows = width // mesh_size, height // mesh_size
    idx = np.arange(mesh_size)
    r = np.arange(radius)
    off_r, off_c = (r * mesh_rows) // 2, (r * mesh_cols) // 2                        # HREM.py:59-60
    sign = np.array([1, -1])
    pi = np.clip(mesh_rows * idx[:, None, None] + sign[None, None, :] * off_r[None, :, None], 0, height - 1)   # (i, r, si)
    pj = np.clip(mesh_cols * idx[:, None, None] + sign[None, None, :] * off_c[None, :, None], 0, width - 1)    # (j, r, sj)
    ii = pi[:, None, :, :, None]                                                     # (i, 1, r, si, 1)
    jj = pj[None, :, :, None, :]                                                     # (1, j, r, 1, sj)
    n = radius * 4

    def mesh(a):
        s = np.sort(a[ii, jj].reshape(mesh_size, mesh_size, n).astype(float), axis=-1)
        m = s[..., n // 2]                                                           # sorted[len // 2]: the upper median
        p = np.pad(m, 2, mode="edge")
        win = np.lib.stride_tricks.sliding_window_view(p, (5, 5)).reshape(mesh_size, mesh_size, 25)
        return np.sort(win, axis=-1)[..., 12]
    return mesh(u), mesh(v)


# ------------------------------------------------------------------------------------------------ dataset
class HREMEventFlow(torch.utils.data.Dataset):
    """HREM event-pair dataset (loader/HREM.py:128-274) with GPU voxelization.

    args: {'eval_type', 'event_interval' ('dt1' | 'dt4'), 'num_voxel_bins'} as in the reference's config; `root`
    replaces the reference's hard-wired repository path: <root>/dataset/HREM/{train,test}/<dt>/[<sequence>/]<sample>/
    {events1.npz, events2.npz, flow.flo}."""

    image_width = 1280
    image_height = 720

    def __init__(self, args, train=True, root=None, device=None, to_cpu=False, augmentor=None, deferred_norm=False):
        """deferred_norm (evaluation, GPU-resident samples): the event volumes stay RAW with their normalisation record behind them
        (voxelizer normalize='deferred') for a model that normalises as it reads - EEMFlow.forward_many(..., deferred_norm=True),
        which harness.TestRaftEvents.test_multi_sequence(coalesce=...) calls when the samples say so (`sample['deferred_norm']`)."""
        super().__init__()
        if deferred_norm and (train or to_cpu):
            raise ValueError("deferred_norm is the evaluation route with samples resident on the GPU (train=False, to_cpu=False)")
        self.deferred_norm = bool(deferred_norm)
        self.input_type = 'events'
        self.type = 'train' if train else 'val'
        self.evaluation_type = args['eval_type']
        self.dt = args['event_interval']
        self.num_bins = args['num_voxel_bins']
        self.root = root if root is not None else os.environ.get("EEMFLOW_DATA_ROOT", os.getcwd())
        self.device = torch.device(device if device is not None else "cuda:0")
        self.to_cpu = to_cpu
        if 'aug_params' in args and augmentor is None and train:
            from .augmentor import FlowAugmentor                  # HREM.py:146-150
            augmentor = FlowAugmentor(**args['aug_params'])
        self.augmentor = augmentor
        self.voxel = EventSequenceToVoxelGrid_Pytorch(num_bins=self.num_bins, normalize="deferred" if deferred_norm else True, gpu=True,
                                                      gpu_nr=self.device.index or 0, forkserver=False)
        self.get_data_ls()

    @staticmethod
    def _samples(folder):
        out = []
        for names in sorted(os.listdir(folder)):
            e1, e2 = os.path.join(folder, names, "events1.npz"), os.path.join(folder, names, "events2.npz")
            if os.path.exists(e1) and os.path.exists(e2):
                out.append({"names": names, "event0": e1, "event1": e2, "fflow": os.path.join(folder, names, "flow.flo")})
        return out

    def get_data_ls(self):
        if self.type == 'train':
            self.dataset_dir = os.path.join(self.root, 'dataset/HREM/train/{:s}'.format(self.dt))
            self.data_ls = self._samples(self.dataset_dir)
        else:
            self.dataset_dir = os.path.join(self.root, 'dataset/HREM/test/{:s}'.format(self.dt))
            self.nori_list = {seq: self._samples(os.path.join(self.dataset_dir, seq)) for seq in sorted(os.listdir(self.dataset_dir))}
            self.data_ls = []

    def change_test_sequence(self, sequence):
        self.data_ls = self.nori_list[sequence]

    def __len__(self):
        return len(self.data_ls)

    def _read(self, idx):
        """Everything of sample idx that comes from files: the dict without its volumes, and its two event sequences."""
        sample = self.data_ls[idx]
        fflow = read_flo(sample['fflow'])
        height, width = fflow.shape[0], fflow.shape[1]
        x_mesh, y_mesh = motion_propagate(fflow, height, width)
        out = {'names': sample["names"],
               'flow': torch.from_numpy(np.stack([x_mesh, y_mesh], axis=0)),
               'fflow': torch.from_numpy(np.ascontiguousarray(fflow.transpose(2, 0, 1))),
               'valid': None}
        params = {'height': self.image_height, 'width': self.image_width}
        seqs = [EventSequence(None, params, features=get_compressed_events(sample[key]), timestamp_multiplier=1e6,
                              convert_to_relative=True) for key in ('event0', 'event1')]
        return out, seqs

    def _attach(self, out, old, new):
        if self.to_cpu:
            old, new = old.cpu(), new.cpu()
        out['event_volume_old'], out['event_volume_new'] = old, new
        if self.deferred_norm:
            # event_valid is the bin sum of the NORMALISED volume (HREM.py:232): from the raw grid and its record
            from .voxelizer import norm_record
            rec = norm_record(old)
            nz = (old != 0).sum(dim=0)
            ev = (old.sum(dim=0) - nz * rec[0]) / rec[1]
            out['event_valid'] = torch.where(rec[3] != 0, ev, old.sum(dim=0)).unsqueeze(0)
            out['deferred_norm'] = True
        else:
            out['event_valid'] = old.sum(dim=0).unsqueeze(0)
        return out

    def get_sample(self, idx):
        out, seqs = self._read(idx)
        old, new = self.voxel.pair(seqs[0], seqs[1])              # both volumes in one three-launch sequence
        return self._attach(out, old, new)

    def get_samples(self, idxs):
        """`[self[i] for i in idxs]` (up to 16 samples) with ALL their volumes from one voxelizer launch sequence."""
        read = [self._read(i % len(self)) for i in idxs]
        vols = self.voxel.many([s for _, seqs in read for s in seqs])
        return [self._finish(self._attach(out, vols[2 * k], vols[2 * k + 1])) for k, (out, _) in enumerate(read)]

    def __getitem__(self, idx):
        return self._finish(self.get_sample(idx % len(self)))

    def _finish(self, sample):
        if self.type == 'train':
            if self.augmentor is not None:
                img1 = sample['event_volume_old'].permute(1, 2, 0).cpu().numpy()
                img2 = sample['event_volume_new'].permute(1, 2, 0).cpu().numpy()
                meshflow = sample['flow'].permute(1, 2, 0).numpy()
                img1, img2, _ = self.augmentor(img1, img2, meshflow, without_resize=True)
                sample['event_volume_old'] = torch.from_numpy(img1).permute(2, 0, 1).float()
                sample['event_volume_new'] = torch.from_numpy(img2).permute(2, 0, 1).float()
            sample['flow'] = sample['flow'].float()               # the un-augmented mesh flow, as the reference returns
            sample['valid'] = torch.ones(size=tuple(sample['flow'].shape[1:]))
        else:
            # mesh flow -> full resolution, bilinear, align_corners=False (HREM.py:264-267), on the GPU
            mesh = sample['flow'].float().to(self.device).contiguous()
            full = torch.empty(2, self.image_height, self.image_width, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().eemflow_upsample_bilinear(mesh.data_ptr(), full.data_ptr(), 2, mesh.shape[1], mesh.shape[2],
                                                                self.image_height, self.image_width,
                                                                _lib.current_stream_ptr(self.device)))
            valid = (~torch.isinf(full[0])) & (~torch.isinf(full[1])) & (torch.linalg.norm(full, dim=0) > 0)
            sample['flow'] = full.cpu() if self.to_cpu else full
            sample['valid'] = valid.float().cpu() if self.to_cpu else valid.float()
        return sample
