"""Wav2Letter on the HIP step engine (reference: wav2letter.py:12-92)."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .base_asr_models import ConvCTCASR
from .engine import ACT_CLAMP20, ACT_NONE, PAD_REFLECT, StackEngine, UnitSpec
from .layers import BatchNorm1d, Conv1d, conv_spec, default_precision, run_stack


class Conv1dBlock(nn.Module):
    """reflect-pad -> Conv1d -> BatchNorm1d(momentum=.9, eps=1e-3) -> Dropout -> clamp(0, 20)
    (wav2letter.py:12-47).  Note the reference's pad rule derives the SAME-padding amount from
    ``input_channels`` (wav2letter.py:24-27); reproduced as is."""

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
        self.padding = kernel_size[0]
        input_rows = input_channels
        filter_rows = kernel_size[0]
        out_rows = (input_rows + stride - 1) // stride
        self.padding_rows = max(0, (out_rows - 1) * stride + (filter_rows - 1) * dilation + 1 - input_rows)
        self.pad_l = self.padding_rows // 2
        self.pad_r = (self.padding_rows + 1) // 2 if self.padding_rows % 2 else self.padding_rows // 2
        self.conv1 = Conv1d(input_channels, output_channels, kernel_size, stride=stride, padding=0, dilation=dilation)
        self.batch_norm = BatchNorm1d(output_channels, momentum=0.9, eps=0.001) if bn else None
        self.has_dropout = self.drop_out_prob != -1

    def unit(self, src: int, name: str = '') -> UnitSpec:
        spec = conv_spec(self.conv1, self.batch_norm, self.pad_l, self.pad_r, PAD_REFLECT, name)
        p = float(self.drop_out_prob) if self.has_dropout else 0.0
        return UnitSpec(main=spec, src=src, act=ACT_CLAMP20 if self.activation_use else ACT_NONE, drop_p=max(p, 0.0))

    def forward(self, xs):
        raise RuntimeError('Conv1dBlock runs inside Wav2Letter.forward on the HIP step engine; '
                           'a stand-alone block has no execution path of its own')


class Wav2Letter(ConvCTCASR):
    def __init__(self, cfg):
        super(Wav2Letter, self).__init__(cfg)
        self.mid_layers = cfg.mid_layers
        if not cfg.input_size:
            nfft = (self.audio_conf['sample_rate'] * self.audio_conf['window_size'])
            self.input_size = int(1 + (nfft / 2))
        else:
            self.input_size = cfg.input_size
        self.precision = default_precision(cfg)

        layers = cfg.layers[: self.mid_layers]
        layer_size = self.input_size
        conv_blocks = []
        for idx in range(len(layers)):
            layer_params = layers[idx]
            layer = Conv1dBlock(input_channels=layer_size, output_channels=layer_params.output_size,
                                kernel_size=(layer_params.kernel_size,), stride=layer_params.stride,
                                dilation=layer_params.dilation, drop_out_prob=layer_params.dropout)
            layer_size = layer_params.output_size
            conv_blocks.append(('conv1d_{}'.format(idx), layer))
        last_layer = Conv1dBlock(input_channels=layer_size, output_channels=len(self.labels), kernel_size=(1,),
                                 stride=1, bn=False, activation_use=False)
        conv_blocks.append(('conv1d_{}'.format(len(layers)), last_layer))
        self.conv1ds = nn.Sequential(OrderedDict(conv_blocks))
        self._engine = None

    @property
    def scaling_factor(self):
        if not hasattr(self, '_scaling_factor'):
            strides = []
            for module in self.conv1ds.children():
                strides.append(module.conv1.stride[0])
            self._scaling_factor = int(np.prod(strides))
        return self._scaling_factor

    def engine(self) -> StackEngine:
        # rebuilt per call: module.to()/cuda() replaces buffer tensors, so specs must not go stale
        precise = self.precision == 'fp32'
        if True:
            blocks = list(self.conv1ds.children())
            units = [b.unit(i, f'conv1d_{i}') for i, b in enumerate(blocks[:-1])]
            head_blk = blocks[-1]
            if head_blk.batch_norm is not None or head_blk.activation_use or head_blk.padding_rows:
                raise NotImplementedError('the classifier block must be a plain 1x1 convolution (wav2letter.py:69)')
            head = conv_spec(head_blk.conv1, None, 0, 0, PAD_REFLECT, 'head')
            self._engine = StackEngine(units, head, len(self.labels), precise=precise)
            self._engine.overlap_wgrad = getattr(self, '_overlap_wgrad', True)
            self._engine.dropout_counter = getattr(self, '_dropout_counter', None)      # graph.GraphedTrainStep
            reducer = getattr(self, 'grad_reducer', None)      # set by distributed training drivers
            if reducer is not None:
                self._engine.grad_ready = reducer.on_grad
                self._engine.flat_ready = getattr(reducer, 'on_flat', None)
                self._engine.backward_done = reducer.finish
        return self._engine

    def forward(self, x, input_lengths=None):
        """x [N, input_size, T] float -> (log_probs [N, T', n_labels], output_lengths or None)
        (wav2letter.py:84-92).  Lengths are not used inside the network (no masking)."""
        if getattr(self, '_debug_keep_ctx', False):      # test hook: expose the engine's saved activations
            x, _, self._last_ctx = run_stack(self.engine(), x, None, self.training, softmax_mode=0, keep_ctx=True)
        else:
            x, _ = run_stack(self.engine(), x, None, self.training, softmax_mode=0)
        if input_lengths is not None:
            output_lengths = self.compute_output_lengths(input_lengths)
        else:
            output_lengths = None
        return x, output_lengths
