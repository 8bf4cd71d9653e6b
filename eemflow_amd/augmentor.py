"""The two augmentors the dataset front-ends use, as they behave in the reference as shipped (utils/augumentor.py).

`FlowAugmentor` as loader/HREM.py:148,252 calls it (`without_resize=True`, :202-257) and `DenseSparseAugmentor` (:329-433, the
one loader/MVSEC.py:57,176 means to use): on these paths every rescaling branch of the reference is commented out and the
eraser / colour jitter are never called, so what runs is random flips and - for DenseSparseAugmentor - a random crop, all on
HWC numpy arrays.  (`FlowAugmentor` with rescaling goes through cv2.resize and raises here.)  The random draws are made with `numpy.random` in the reference's order, so a seeded run reproduces the reference's
output bit for bit (tests/golden/augmentor.npz is produced by executing the reference classes).  cv2 / torchvision, which
the reference imports for the dead branches, are not needed.
"""
import numpy as np


class FlowAugmentor:
    def __init__(self, crop_size, min_scale=-0.2, max_scale=0.5, do_flip=False):
        self.crop_size = crop_size
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.do_flip = do_flip
        self.h_flip_prob = 0.5
        self.v_flip_prob = 0.1

    def spatial_transform(self, img1, img2, flow):
        # augumentor.py:158-200: random rescale through cv2.resize(INTER_LINEAR), flips, crop.  No loader of the path calls it
        # (HREM.py:252 passes without_resize=True) and cv2 is absent here, so its interpolation cannot be pinned: not built.
        raise NotImplementedError("FlowAugmentor with rescaling (cv2.resize) is not built; call with without_resize=True")

    def spatial_transform_no_resize(self, img1, img2, flow):
        if self.do_flip:                                           # :225-234
            if np.random.rand() < self.h_flip_prob:
                img1 = img1[:, ::-1]
                img2 = img2[:, ::-1]
                flow = flow[:, ::-1] * [-1.0, 1.0]
            if np.random.rand() < self.v_flip_prob:
                img1 = img1[::-1, :]
                img2 = img2[::-1, :]
                flow = flow[::-1, :] * [1.0, -1.0]
        return img1, img2, flow

    def __call__(self, img1, img2, flow, without_resize=False):
        if without_resize:
            img1, img2, flow = self.spatial_transform_no_resize(img1, img2, flow)
        else:
            img1, img2, flow = self.spatial_transform(img1, img2, flow)
        return np.ascontiguousarray(img1), np.ascontiguousarray(img2), np.ascontiguousarray(flow)


class DenseSparseAugmentor:
    def __init__(self, crop_size, min_scale=-0.2, max_scale=0.5, do_flip=False):
        self.crop_size = crop_size
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.do_flip = do_flip
        self.h_flip_prob = 0.5
        self.v_flip_prob = 0.1

    def spatial_transform(self, img1, img2, dimg1, dimg2, flow):
        if self.do_flip:                                           # :389-403
            if np.random.rand() < self.h_flip_prob:
                img1, img2, dimg1, dimg2 = img1[:, ::-1], img2[:, ::-1], dimg1[:, ::-1], dimg2[:, ::-1]
                flow = flow[:, ::-1] * [-1.0, 1.0]
            if np.random.rand() < self.v_flip_prob:
                img1, img2, dimg1, dimg2 = img1[::-1, :], img2[::-1, :], dimg1[::-1, :], dimg2[::-1, :]
                flow = flow[::-1, :] * [1.0, -1.0]
        ch, cw = self.crop_size                                    # :405-419
        y0 = 0 if img1.shape[0] == ch else np.random.randint(0, img1.shape[0] - ch)
        x0 = 0 if img1.shape[1] == cw else np.random.randint(0, img1.shape[1] - cw)
        sl = (slice(y0, y0 + ch), slice(x0, x0 + cw))
        return img1[sl], img2[sl], dimg1[sl], dimg2[sl], flow[sl]

    def __call__(self, img1, img2, dimg1, dimg2, flow):
        out = self.spatial_transform(img1, img2, dimg1, dimg2, flow)
        return tuple(np.ascontiguousarray(a) for a in out)
