"""InputPadder with the reference's interface (utils/image_utils.py:126-145): constructor arguments, the `_pad` list in F.pad order
[left, right, top, bottom], `pad(*tensors)` and `unpad(tensor)`.

Inside EEMFlow the padding is folded into the first HIP conv kernel; this class exists for harness code that pads / unpads explicitly
(E-RAFT style callers) and to expose the pad arithmetic (pinned by tests/golden/pad.npz, generated from the reference class)."""
import torch.nn.functional as F


def _split(total, centred):
    """A pad of `total` pixels as (before, after): halved with the odd pixel after, or all of it after."""
    before = total // 2 if centred else 0
    return before, total - before


class InputPadder:
    """Replicate-pads the last two dimensions up to the next multiple of `eval_pad_rate`."""

    def __init__(self, dims, mode='sintel', eval_pad_rate=32):
        self.eval_pad_rate = eval_pad_rate
        self.ht, self.wd = dims[-2:]
        missing_h = -self.ht % eval_pad_rate              # pixels up to the next multiple (0 when already one)
        missing_w = -self.wd % eval_pad_rate
        left, right = _split(missing_w, True)
        top, bottom = _split(missing_h, mode == 'sintel')  # other modes (kitti) put the rows at the bottom
        self._pad = [left, right, top, bottom]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def unpad(self, x):
        left, right, top, bottom = self._pad
        return x[..., top:x.shape[-2] - bottom, left:x.shape[-1] - right]
