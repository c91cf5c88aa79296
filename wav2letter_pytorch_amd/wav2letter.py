"""Wav2Letter on the HIP step engine (reference: wav2letter.py:12-92)."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Tuple

import torch.nn as nn

from .base_asr_models import ConvCTCASR, feature_size
from .engine import ACT_CLAMP20, ACT_NONE, PAD_REFLECT, StackEngine, UnitSpec
from .layers import BatchNorm1d, Conv1d, conv_spec, default_precision, run_stack, solo_engine


def same_pad_amounts(rows: int, kernel: int, stride: int, dilation: int) -> Tuple[int, int, int]:
    """TF-style SAME padding of an axis of ``rows`` samples: (total, left, right), the odd sample on the right.
    The reference feeds this rule the number of input CHANNELS, not the sequence length (wav2letter.py:24-27); for
    stride 1 the result is length-independent ((k-1)*d), and the 64-mel stride-2 first layer gets 9 = (4, 5)."""
    covered = (math.ceil(rows / stride) - 1) * stride + (kernel - 1) * dilation + 1
    total = max(0, covered - rows)
    return total, total // 2, total - total // 2


class Conv1dBlock(nn.Module):
    """reflect-pad -> Conv1d -> BatchNorm1d(momentum=.9, eps=1e-3) -> Dropout -> clamp(0, 20)
    (wav2letter.py:12-47).  Inside Wav2Letter the block is one unit of the model's step engine; called on its own
    (``block(x)`` with x [N, C_in, T] on the device) it runs as a one-unit engine with autograd."""

    def __init__(self, input_channels, output_channels, kernel_size, stride, drop_out_prob=-1.0, dilation=1, bn=True,
                 activation_use=True):
        super(Conv1dBlock, self).__init__()
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.drop_out_prob = drop_out_prob
        self.dilation = dilation
        self.activation_use = activation_use
        self.padding = kernel_size[0]                      # kept (and unused) as in the reference, wav2letter.py:22
        self.padding_rows, self.pad_l, self.pad_r = same_pad_amounts(input_channels, kernel_size[0], stride, dilation)
        self.conv1 = Conv1d(input_channels, output_channels, kernel_size, stride=stride, padding=0, dilation=dilation)
        self.batch_norm = BatchNorm1d(output_channels, momentum=0.9, eps=0.001) if bn else None
        self.has_dropout = self.drop_out_prob != -1
        self.precision = default_precision()

    def unit(self, src: int, name: str = '') -> UnitSpec:
        spec = conv_spec(self.conv1, self.batch_norm, self.pad_l, self.pad_r, PAD_REFLECT, name)
        p = float(self.drop_out_prob) if self.has_dropout else 0.0
        return UnitSpec(main=spec, src=src, act=ACT_CLAMP20 if self.activation_use else ACT_NONE, drop_p=max(p, 0.0))

    def forward(self, xs):
        """[N, C_in, T] -> [N, C_out, T'] fp32 (wav2letter.py:40-47)"""
        eng = solo_engine(self, lambda: [self.unit(0, 'conv1d')])
        if getattr(self, '_debug_keep_ctx', False):        # test hook: expose the engine's saved activations
            out, _, self._last_ctx = run_stack(eng, xs, None, self.training, keep_ctx=True)
            return out
        out, _ = run_stack(eng, xs, None, self.training)
        return out


class Wav2Letter(ConvCTCASR):
    def __init__(self, cfg):
        super(Wav2Letter, self).__init__(cfg)
        self.mid_layers = cfg.mid_layers
        self.input_size = feature_size(cfg, self.audio_conf)
        self.precision = default_precision(cfg)
        rows = list(cfg.layers[: self.mid_layers])
        widths = [self.input_size] + [r.output_size for r in rows]
        blocks = [Conv1dBlock(input_channels=c_in, output_channels=r.output_size, kernel_size=(r.kernel_size,),
                              stride=r.stride, dilation=r.dilation, drop_out_prob=r.dropout)
                  for c_in, r in zip(widths, rows)]
        # classifier: 1x1 conv to the label set, no BatchNorm / activation / dropout (wav2letter.py:69-70)
        blocks.append(Conv1dBlock(input_channels=widths[-1], output_channels=len(self.labels), kernel_size=(1,), stride=1,
                                  bn=False, activation_use=False))
        self.conv1ds = nn.Sequential(OrderedDict((f'conv1d_{i}', b) for i, b in enumerate(blocks)))

    @property
    def scaling_factor(self):
        """input frames per output frame = product of the conv strides (wav2letter.py:74-81), computed once"""
        cached = self.__dict__.get('_scaling_factor')
        if cached is None:
            cached = self.__dict__['_scaling_factor'] = math.prod(b.conv1.stride[0] for b in self.conv1ds.children())
        return cached

    def _build_engine(self) -> StackEngine:
        *body, head_blk = self.conv1ds.children()
        if head_blk.batch_norm is not None or head_blk.activation_use or head_blk.padding_rows:
            raise NotImplementedError('the classifier block must be a plain 1x1 convolution (wav2letter.py:69)')
        units = [b.unit(i, f'conv1d_{i}') for i, b in enumerate(body)]
        head = conv_spec(head_blk.conv1, None, 0, 0, PAD_REFLECT, 'head')
        return StackEngine(units, head, len(self.labels), precise=self.precision == 'fp32', fp8=self.precision == 'fp8')

    def engine(self) -> StackEngine:
        return self._cached_engine(self._build_engine)

    def forward(self, x, input_lengths=None):
        """x [N, input_size, T] float -> (log_probs [N, T', n_labels], output_lengths or None)
        (wav2letter.py:84-92).  Lengths are not used inside the network (no masking)."""
        keep = getattr(self, '_debug_keep_ctx', False)        # test hook: expose the engine's saved activations
        res = run_stack(self.engine(), x, None, self.training, softmax_mode=0, keep_ctx=keep)
        if keep:
            self._last_ctx = res[2]
        return res[0], (None if input_lengths is None else self.compute_output_lengths(input_lengths))
