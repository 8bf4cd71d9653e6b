"""The two augmentors the dataset front-ends use, as they behave in the reference as shipped (utils/augumentor.py).

`FlowAugmentor` as loader/HREM.py:148,252 calls it (`without_resize=True`, :202-257) and `DenseSparseAugmentor` (:329-433, the
one loader/MVSEC.py:57,176 means to use): on these paths every rescaling branch of the reference is commented out and the
eraser / colour jitter are never called, so what runs is random flips and - for DenseSparseAugmentor - a random crop, all on
HWC numpy arrays.  `FlowAugmentor` with rescaling (no loader of the path calls it) goes through cv2.resize(INTER_LINEAR) in the
reference; cv2 is absent here, so `resize_linear` restates its rule for floating-point arrays (half-pixel centres, edge clamp, output
size round(src * scale)) - PARITY UNPINNED against cv2 itself, checked against torch's bilinear interpolation which follows the
same convention (tests/test_data_rows.py).  The random draws are made with `numpy.random` in the reference's order, so a seeded run reproduces the reference's
output bit for bit (tests/golden/augmentor.npz is produced by executing the reference classes).  cv2 / torchvision, which
the reference imports for the dead branches, are not needed.
"""
import numpy as np


def resize_linear(img, fx, fy):
    """cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for a floating-point HWC (or HW) array: output size
    (round(h * fy), round(w * fx)); destination pixel d samples the source at (d + 0.5) / f - 0.5 with f the GIVEN factor (cv2 uses
    fx / fy only to round the output size and maps with 1 / fx, 1 / fy - not with src / dst, which drifts by a sub-pixel towards the far
    edge whenever src * f is not an integer), the two neighbours clamped to the image."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    oh, ow = int(round(h * fy)), int(round(w * fx))
    if oh < 1 or ow < 1:
        raise ValueError("resize_linear: empty output")

    def axis(n_src, n_dst, f):
        s = (np.arange(n_dst, dtype=np.float64) + 0.5) / float(f) - 0.5
        i0 = np.floor(s).astype(np.int64)
        t = s - i0
        lo = np.clip(i0, 0, n_src - 1)
        hi = np.clip(i0 + 1, 0, n_src - 1)
        return lo, hi, t

    y0, y1, ty = axis(h, oh, fy)
    x0, x1, tx = axis(w, ow, fx)
    f = a.astype(np.float64)
    tx = tx.reshape((1, ow) + (1,) * (a.ndim - 2))
    ty = ty.reshape((oh, 1) + (1,) * (a.ndim - 2))
    top = f[y0][:, x0] * (1.0 - tx) + f[y0][:, x1] * tx
    bot = f[y1][:, x0] * (1.0 - tx) + f[y1][:, x1] * tx
    return (top * (1.0 - ty) + bot * ty).astype(a.dtype if a.dtype.kind == "f" else np.float64)


class FlowAugmentor:
    def __init__(self, crop_size, min_scale=-0.2, max_scale=0.5, do_flip=False):
        self.crop_size = crop_size
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.do_flip = do_flip
        self.h_flip_prob = 0.5
        self.v_flip_prob = 0.1
        self.spatial_aug_prob = 0.8                                # utils/augumentor.py:23-25
        self.stretch_prob = 0.8
        self.max_stretch = 0.2

    def spatial_transform(self, img1, img2, flow):
        """utils/augumentor.py:158-200: random rescale (the draws in the reference's order), flips, random crop."""
        ht, wd = img1.shape[:2]
        min_scale = np.maximum((self.crop_size[0] + 8) / float(ht), (self.crop_size[1] + 8) / float(wd))
        scale = 2 ** np.random.uniform(self.min_scale, self.max_scale)
        scale_x = scale_y = scale
        if np.random.rand() < self.stretch_prob:
            scale_x *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
            scale_y *= 2 ** np.random.uniform(-self.max_stretch, self.max_stretch)
        scale_x = np.clip(scale_x, min_scale, None)
        scale_y = np.clip(scale_y, min_scale, None)
        if np.random.rand() < self.spatial_aug_prob:
            img1 = resize_linear(img1, scale_x, scale_y)
            img2 = resize_linear(img2, scale_x, scale_y)
            flow = resize_linear(flow, scale_x, scale_y) * [scale_x, scale_y]
        if self.do_flip:
            if np.random.rand() < self.h_flip_prob:
                img1, img2 = img1[:, ::-1], img2[:, ::-1]
                flow = flow[:, ::-1] * [-1.0, 1.0]
            if np.random.rand() < self.v_flip_prob:
                img1, img2 = img1[::-1, :], img2[::-1, :]
                flow = flow[::-1, :] * [1.0, -1.0]
        y0 = np.random.randint(0, img1.shape[0] - self.crop_size[0])
        x0 = np.random.randint(0, img1.shape[1] - self.crop_size[1])
        ch, cw = self.crop_size[0], self.crop_size[1]
        return img1[y0:y0 + ch, x0:x0 + cw], img2[y0:y0 + ch, x0:x0 + cw], flow[y0:y0 + ch, x0:x0 + cw]

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
