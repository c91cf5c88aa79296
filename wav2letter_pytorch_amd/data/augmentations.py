"""SpecAugment / SpecCutout / Identity with the reference's constructor arguments and random draws
(data/augmentations.py:11-106).  The rectangles are drawn on the host from the module's ``random.Random`` in the
reference's order, so a seeded rng cuts exactly the same cells; the fill itself is one HIP launch (w2l_zero_rects)
on the device-resident batch.  Input: float32 CUDA tensor [N, F, T]; returns a new tensor (masked_fill semantics)."""
import random

import torch
import torch.nn as nn

from .._lib import check, lib, ptr, require_device, stream_ptr


def _cut(x: torch.Tensor, rects) -> torch.Tensor:
    require_device(x)
    if x.dim() != 3:
        raise ValueError('expected a [N, F, T] batch of spectrograms')
    out = x.detach().to(torch.float32).contiguous().clone()
    n, f, t = out.shape
    norm = []
    for (i, f0, f1, t0, t1) in rects:                     # Python slice semantics of mask[i, f0:f1, t0:t1] = 1
        fa, fb, _ = slice(f0, f1).indices(f)
        ta, tb, _ = slice(t0, t1).indices(t)
        if fb > fa and tb > ta:
            norm.append((i, fa, fb, ta, tb))
    if norm:
        r = torch.tensor(norm, dtype=torch.int32).to(out.device, non_blocking=True)
        check(lib.w2l_zero_rects(ptr(out), n, f, t, ptr(r), len(norm), stream_ptr()), 'w2l_zero_rects')
    return out


class SpecAugment(nn.Module):
    """Zeroes ``freq_masks`` horizontal and ``time_masks`` vertical bands per spectrogram (augmentations.py:11-58)."""

    def __init__(self, freq_masks=1, time_masks=1, freq_width=15, time_width=50, rng=None):
        super().__init__()
        self._rng = random.Random() if rng is None else rng
        self.freq_masks, self.time_masks = freq_masks, time_masks
        self.freq_width, self.time_width = freq_width, time_width

    def rectangles(self, shape):
        n, f, t = shape
        rects = []
        for i in range(n):
            for _ in range(self.freq_masks):
                left = int(self._rng.uniform(0, f - self.freq_width))
                w = int(self._rng.uniform(0, self.freq_width))
                rects.append((i, left, left + w, None, None))
            for _ in range(self.time_masks):
                left = int(self._rng.uniform(0, t - self.time_width))
                w = int(self._rng.uniform(0, self.time_width))
                rects.append((i, None, None, left, left + w))
        return rects

    @torch.no_grad()
    def forward(self, x):
        return _cut(x, self.rectangles(x.shape))


class SpecCutout(nn.Module):
    """Zeroes ``rect_masks`` rectangles per spectrogram (augmentations.py:61-99).  As in the reference, the extent along
    the frequency axis is drawn from ``rect_time`` and the extent along time from ``rect_freq``."""

    def __init__(self, rect_masks=5, rect_time=60, rect_freq=25, rng=None):
        super().__init__()
        self._rng = random.Random() if rng is None else rng
        self.rect_masks, self.rect_time, self.rect_freq = rect_masks, rect_time, rect_freq

    def rectangles(self, shape):
        n, f, t = shape
        rects = []
        for i in range(n):
            for _ in range(self.rect_masks):
                rf = int(self._rng.uniform(0, f - self.rect_freq))
                rt = int(self._rng.uniform(0, t - self.rect_time))
                wf = int(self._rng.uniform(0, self.rect_time))
                wt = int(self._rng.uniform(0, self.rect_freq))
                rects.append((i, rf, rf + wf, rt, rt + wt))
        return rects

    @torch.no_grad()
    def forward(self, x):
        return _cut(x, self.rectangles(x.shape))


class Identity(nn.Module):
    @torch.no_grad()
    def forward(self, x):
        return x
