"""MVSEC event-pair datasets feeding the GPU voxelizer (reference: loader/MVSEC.py:23-283, loader_utils.py:41-51).

`MvsecEventFlow` (dt1) and `MvsecEventFlow_dt4` keep the reference's args dict, directory layout, sequence frame ranges
and sample keys; `root` replaces its hard-wired repository path:

    <root>/dataset/MVSEC/<sequence>/event/%06d.h5          per-frame pandas tables 'myDataset' with ts, x, y, p
    <root>/dataset/MVSEC/<sequence>/flowgt_dt{1,4}/%d.npy  ground-truth flow, (2,H,W) or (H,W,2)
    <root>/dataset/MVSEC_test/indoor_flying1/...           the dt1 exception of MVSEC.py:75-78

Event files: the reference reads them with pandas.read_hdf (PyTables).  That is still the default reader here and it
fails loudly where PyTables is missing; `events_reader=callable(path) -> (N,4) float64 [ts, x, y, p]` replaces it, and a
`%06d.npz` with arrays ts, x, y, p next to (or instead of) the .h5 is read directly (the exported form for machines
without PyTables).

Differences, all deliberate:
  * the voxel volumes stay on the GPU unless `to_cpu=True` (the reference voxelizes on the GPU and copies back);
  * the shipped module cannot be imported as is: it takes FlowAugmentor / DenseSparseAugmentor from loader_utils, which
    does not define them (MVSEC.py:20); they are in utils/augumentor.py and restated in eemflow_amd/augmentor.py.  With
    'aug_params' in args a training sample goes through DenseSparseAugmentor (flips + random crop, MVSEC.py:170-187);
    without it samples are returned un-augmented (the reference would crash there), or through a caller-supplied
    `augmentor=callable(event1, event2, flow) -> (event1, event2, flow)` (HWC numpy, MVSEC.py:183);
  * dt4: the reference's `events0.sort_values(by=['ts'])` discards its result (MVSEC.py:256-258); what orders the
    concatenated events is EventSequence's own sort, as here.
"""
import os

import numpy as np
import torch

from .voxelizer import EventSequence, EventSequenceToVoxelGrid_Pytorch

Valid_Time_Index = {                                  # MVSEC.py:23-30
    'indoor_flying1': [(314, 2199)],
    'indoor_flying2': [(314, 2199)],
    'indoor_flying3': [(314, 2199)],
    'indoor_flying4': [(196, 570)],
    'outdoor_day1': [(245, 3000)],
    'outdoor_day2': [(4375, 7002)],
}


def get_events(event_path):
    """(N,4) float64 [ts, x, y, p] of one frame's events (loader_utils.py:41-51).  `<stem>.npz` (arrays ts, x, y, p) is
    preferred when present; otherwise the pandas/PyTables table 'myDataset'."""
    npz = os.path.splitext(event_path)[0] + ".npz"
    if os.path.exists(npz):
        with np.load(npz) as f:
            return np.stack([f[k].astype(np.float64) for k in ("ts", "x", "y", "p")], axis=1)
    if not os.path.exists(event_path):
        raise FileNotFoundError(event_path)           # the reference prints and returns 0, which then fails later
    try:
        import pandas
        frame = pandas.read_hdf(event_path, "myDataset")
    except ImportError as e:                          # PyTables missing
        raise RuntimeError(f"reading {event_path} needs pandas + PyTables ({e}); export the table to "
                           f"{npz} (arrays ts, x, y, p) or pass events_reader=") from e
    return frame[['ts', 'x', 'y', 'p']].to_numpy().astype(np.float64)


def center_crop(x, size):
    """torchvision.transforms.CenterCrop((th, tw)) on the last two dims (MVSEC.py:52,193-197): top = round((H-th)/2),
    left = round((W-tw)/2), Python round (banker's) as torchvision uses."""
    th, tw = size
    h, w = x.shape[-2:]
    if h < th or w < tw:
        raise ValueError(f"center_crop: {h}x{w} is smaller than {th}x{tw}")
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return x[..., top:top + th, left:left + tw]


def event_mask(features, height, width):
    """`np.histogram2d(x, y, bins=(w,h), range=[[0,w],[0,h]]).T > 0` of MVSEC.py:143-151 for integer-valued or
    fractional coordinates: pixel (y, x) holds floor(x), floor(y) for 0 <= x < w (x == w falls into the last bin)."""
    x, y = features[:, 1], features[:, 2]
    ok = (x >= 0) & (x <= width) & (y >= 0) & (y <= height)
    xi = np.minimum(np.floor(x[ok]).astype(np.int64), width - 1)
    yi = np.minimum(np.floor(y[ok]).astype(np.int64), height - 1)
    mask = np.zeros(height * width, dtype=bool)
    mask[yi * width + xi] = True
    return mask.reshape(height, width)


class MvsecEventFlow(torch.utils.data.Dataset):
    """dt1 pairs: events of frame i+1 (old) and i+2 (new) with flowgt_dt1/i.npy (MVSEC.py:32-199)."""

    image_width = 346
    image_height = 260
    frames_per_volume = 1
    flow_dir = 'flowgt_dt1'

    def __init__(self, args, train=True, root=None, device=None, to_cpu=False, augmentor=None, events_reader=None,
                 valid_time_index=None):
        super().__init__()
        self.input_type = 'events'
        self.type = 'train' if train else 'val'
        self.evaluation_type = args['eval_type']
        self.root = root if root is not None else os.environ.get("EEMFLOW_DATA_ROOT", os.getcwd())
        self.device = torch.device(device if device is not None else "cuda:0")
        self.to_cpu = to_cpu
        self.read_events = events_reader or get_events
        self.valid_time_index = valid_time_index or Valid_Time_Index
        self.num_bins = args['num_voxel_bins']
        self.voxel = EventSequenceToVoxelGrid_Pytorch(num_bins=self.num_bins, normalize=True, gpu=True,
                                                      gpu_nr=self.device.index or 0, forkserver=False)
        self.dense_augmentor = None
        if 'aug_params' in args and augmentor is None and train:
            # MVSEC.py:54-57 (which imports the classes from loader_utils, where they are not defined; they live in
            # utils/augumentor.py): samples carry d_event_volume_* keys, so __getitem__ takes the dense-sparse one (:170-176)
            from .augmentor import DenseSparseAugmentor
            self.dense_augmentor = DenseSparseAugmentor(**args['aug_params'])
        self.augmentor = augmentor
        self.change_test_sequence(args['sequence'])

    # ---------------------------------------------------------------------------------------- file lists
    def _dirs(self, sequence):
        if self.frames_per_volume == 1 and sequence == 'indoor_flying1':              # MVSEC.py:75-78
            base = os.path.join(self.root, 'dataset/MVSEC_test', sequence)
        else:
            base = os.path.join(self.root, 'dataset/MVSEC', sequence)
        return os.path.join(base, self.flow_dir), os.path.join(base, 'event')

    def change_test_sequence(self, sequence):
        self.names = [ind for s in self.valid_time_index[sequence] for ind in range(s[0], s[1])]
        self.sequence = 'outdoor_day1' if 'outdoor_day1' in sequence else sequence
        self.flowgt_path, self.event_path = self._dirs(self.sequence)
        if self.frames_per_volume == 1 and self.sequence == 'indoor_flying1':
            self.sequence = 'indoor_flying1_new'
        self.flow_list = [os.path.join(self.flowgt_path, '{:d}.npy'.format(i)) for i in self.names]
        self.event_list = [os.path.join(self.event_path, '{:06d}.h5'.format(i + 1)) for i in self.names]
        last = self.names[-1]
        extra = 1 if self.frames_per_volume == 1 else 5                                # MVSEC.py:93 / :226-227
        self.event_list += [os.path.join(self.event_path, '{:06d}.h5'.format(last + 2 + j)) for j in range(extra)]

    def summary(self, logger):
        logger.write_line("================================== Dataloader Summary ====================================", True)
        logger.write_line("Loader Type:\t\t" + self.__class__.__name__ + " for {}".format(self.type), True)

    def __len__(self):
        return len(self.names)

    # ---------------------------------------------------------------------------------------- one sample
    def _sequence(self, paths):
        feats = np.concatenate([np.asarray(self.read_events(p), dtype=np.float64) for p in paths], axis=0)
        return EventSequence(None, {'height': self.image_height, 'width': self.image_width}, features=feats,
                             timestamp_multiplier=1e6, convert_to_relative=True)

    def get_sample(self, idx):
        flow = np.load(self.flow_list[idx])
        if flow.shape[-1] == 2:
            flow = flow.transpose(2, 0, 1)
        out = {'idx': self.names[idx], 'flow': torch.from_numpy(np.ascontiguousarray(flow)), 'valid': None}
        k = self.frames_per_volume
        n = len(self.event_list)
        old = self._sequence([self.event_list[idx + i] for i in range(k)])                      # MVSEC.py:119,247
        new = self._sequence([self.event_list[(idx + i + 1) % n] for i in range(k)])            # :120,251
        vol_old, vol_new = self.voxel.pair(old, new)              # both volumes in one three-launch sequence
        if self.to_cpu:
            vol_new, vol_old = vol_new.cpu(), vol_old.cpu()
        out['event_volume_new'] = out['d_event_volume_new'] = vol_new
        out['event_volume_old'] = out['d_event_volume_old'] = vol_old
        if self.type == 'val':
            out['event_valid'] = torch.from_numpy(event_mask(old.get_sequence_only(), self.image_height,
                                                             self.image_width)).unsqueeze(dim=0)
        return out

    def __getitem__(self, idx):
        sample = self.get_sample(idx % len(self))
        if self.type == 'train':
            if self.dense_augmentor is not None:                                                # MVSEC.py:170-187
                e1 = sample['event_volume_old'].permute(1, 2, 0).cpu().numpy()
                e2 = sample['event_volume_new'].permute(1, 2, 0).cpu().numpy()
                fl = sample['flow'].permute(1, 2, 0).numpy()
                e1, e2, d1, d2, fl = self.dense_augmentor(e1, e2, e1, e2, fl)
                for key, arr in (('event_volume_old', e1), ('event_volume_new', e2), ('d_event_volume_old', d1),
                                 ('d_event_volume_new', d2)):
                    sample[key] = torch.from_numpy(arr).permute(2, 0, 1).float()
                sample['flow'] = torch.from_numpy(fl).permute(2, 0, 1)
            elif self.augmentor is not None:
                e1 = sample['event_volume_old'].permute(1, 2, 0).cpu().numpy()
                e2 = sample['event_volume_new'].permute(1, 2, 0).cpu().numpy()
                fl = sample['flow'].permute(1, 2, 0).numpy()
                e1, e2, fl = self.augmentor(e1, e2, fl)
                sample['event_volume_old'] = torch.from_numpy(np.ascontiguousarray(e1)).permute(2, 0, 1).float()
                sample['event_volume_new'] = torch.from_numpy(np.ascontiguousarray(e2)).permute(2, 0, 1).float()
                sample['flow'] = torch.from_numpy(np.ascontiguousarray(fl)).permute(2, 0, 1)
            fl = sample['flow'].float()
            sample['flow'] = fl
            sample['valid'] = (~torch.isinf(fl[0]) & ~torch.isinf(fl[1]) & (torch.linalg.norm(fl, dim=0) > 0)).float()  # :185
        else:
            crop = (256, 256)                                                                   # MVSEC.py:52,193-197
            sample['flow'] = center_crop(sample['flow'], crop)
            sample['valid'] = (sample['flow'][0].abs() < 1000) & (sample['flow'][1].abs() < 1000)
            sample['event_volume_old'] = center_crop(sample['event_volume_old'], crop)
            sample['event_volume_new'] = center_crop(sample['event_volume_new'], crop)
            sample['event_valid'] = center_crop(sample['event_valid'], crop)
        return sample


class MvsecEventFlow_dt4(MvsecEventFlow):
    """dt4: four consecutive event frames per volume, flowgt_dt4/i.npy (MVSEC.py:201-283)."""
    frames_per_volume = 4
    flow_dir = 'flowgt_dt4'
