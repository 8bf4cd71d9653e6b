"""InputPadder with the reference's interface (utils/image_utils.py:126-145).

Inside EEMFlow the padding is folded into the first HIP conv kernel; this class exists for harness
code that pads/unpads explicitly (E-RAFT style callers) and to expose the pad arithmetic."""
import torch.nn.functional as F


class InputPadder:
    """ Pads images such that dimensions are divisible by eval_pad_rate """

    def __init__(self, dims, mode='sintel', eval_pad_rate=32):
        self.eval_pad_rate = eval_pad_rate
        self.ht, self.wd = dims[-2:]
        r = eval_pad_rate
        pad_ht = (((self.ht // r) + 1) * r - self.ht) % r
        pad_wd = (((self.wd // r) + 1) * r - self.wd) % r
        if mode == 'sintel':
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]
