"""Jasper on the HIP step engine (reference: jasper.py:21-475, itself derived from NVIDIA's
DeepLearningExamples Jasper).  Module tree, constructor arguments, state-dict keys
(``jasper_encoder.{b}.mconv.{j}.conv.weight`` / ``.mconv.{j}.{weight,bias,running_*}`` /
``.res.{r}.{j}...`` / ``final_layer.0.{weight,bias}``) and quirks follow the reference:
even kernel sizes are bumped to odd (jasper.py:53-58), lengths are updated with true division
(:109-112), eval mode returns softmax instead of log-softmax (:470-473), NaN assert (:474).

Only what ``Jasper._build_encoder`` can reach is executable (jasper.py:440-449): batch
normalisation, ReLU, 'add' residual from the block input, masked (``conv_mask: True``, the default) or plain
(``conv_mask: False``: bare ``Conv1d`` modules whose keys carry no ``.conv``, no length masking, lengths passed through
unchanged -- jasper.py:288-298,393-397) convolutions, dense or separable; separable blocks run a depthwise kernel
(csrc/dwconv.hip) followed by the 1x1 pointwise implicit GEMM."""
from __future__ import annotations

from typing import List

import math

import torch
import torch.nn as nn

from .base_asr_models import ConvCTCASR, feature_size
from .engine import ACT_NONE, ACT_RELU, PAD_ZERO, StackEngine, UnitSpec
from .layers import BatchNorm1d, Conv1d, conv_spec, default_precision, depthwise_forward, run_stack, solo_engine

jasper_activations = {
    "hardtanh": nn.Hardtanh,
    "relu": nn.ReLU,
    "selu": nn.SELU,
}


def init_weights(m, mode='xavier_uniform'):
    """Module initialiser for ``nn.Module.apply`` (jasper.py:29-50): convolutions (bare or wrapped in a MaskedConv1d) get
    xavier-uniform weights (gain 1) -- the only mode there is --, BatchNorm layers go back to their blank state (zero mean,
    unit variance, gamma 1, beta 0, no batches seen)."""
    conv = m.conv if isinstance(m, MaskedConv1d) else m
    if isinstance(conv, Conv1d):
        if mode != 'xavier_uniform':
            raise ValueError("Unknown Initialization mode: {0}".format(mode))
        conv.reset_parameters('xavier_uniform')
    elif isinstance(m, BatchNorm1d):
        with torch.no_grad():
            for buf, value in ((m.running_mean, 0.0), (m.running_var, 1.0), (m.weight, 1.0), (m.bias, 0.0)):
                buf.fill_(value)
            m.num_batches_tracked.zero_()


def compute_new_kernel_size(kernel_size, kernel_width):
    """kernel taps after scaling by ``kernel_width``, forced odd so that 'same' padding is symmetric (jasper.py:53-58):
    32 -> 33, 38 -> 39, ... ; never below one tap"""
    return max(int(kernel_size * kernel_width), 1) | 1


def get_same_padding(kernel_size, stride, dilation):
    """zeros on each side of the time axis (jasper.py:61-66): half the kernel, or half the dilated span minus one"""
    if min(stride, dilation) > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    return (kernel_size * dilation) // 2 - int(dilation > 1)


# per-block keys of cfg.jasper_blocks -> JasperBlock keyword, with the value used when the key is absent
# (jasper.py:440-449); ``layer_size``, ``kernel_size`` and ``residual`` are mandatory
_BLOCK_KEYS = (('layer_size', 'planes', None), ('kernel_size', 'kernel_size', None), ('residual', 'residual', None),
               ('stride', 'stride', 1), ('dilation', 'dilation', 1), ('repeat', 'repeat', 1),
               ('conv_mask', 'conv_mask', True), ('separable', 'separable', True), ('dropout', 'dropout', 0))


def block_kwargs(row) -> dict:
    """one row of ``cfg.jasper_blocks`` as JasperBlock constructor keywords"""
    return {kw: (row[key] if default is None else row.get(key, default)) for key, kw, default in _BLOCK_KEYS}


class _NoParams(nn.Module):
    """stands where the reference keeps an activation / Dropout module so that ModuleList indices
    (= state-dict keys of the later entries) stay the reference's"""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


class MaskedConv1d(nn.Module):
    """jasper.py:69-132: masked_fill by length + nn.Conv1d(bias=False by default)."""
    __constants__ = ["use_conv_mask", "real_out_channels", "heads"]

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, heads=-1,
                 bias=False, use_mask=True):
        super(MaskedConv1d, self).__init__()
        if not (heads == -1 or groups == in_channels):
            raise ValueError("Only use heads for depthwise convolutions")
        if heads != -1:
            raise NotImplementedError('heads != -1 is not reachable from Jasper._build_encoder (jasper.py:440-449)')
        self.real_out_channels = out_channels
        self.conv = Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                           groups=groups, bias=bias, init='xavier_uniform')
        self.use_mask = use_mask
        self.heads = heads

    def get_seq_len(self, lens):
        return (lens + 2 * self.conv.padding[0] - self.conv.dilation[0] * (self.conv.kernel_size[0] - 1) - 1
                ) / self.conv.stride[0] + 1

    def forward(self, x, lens):
        """(x [N, C, T], lens [N]) -> (conv(masked x), updated float lens) -- jasper.py:114-132 -- as a one-unit open engine
        with autograd; inside Jasper the model's engine runs the conv fused with its neighbours."""
        if self.conv.groups != 1:          # depthwise half of a separable block, on its own (jasper.py:319-330)
            out = depthwise_forward(self.conv, x, lens if self.use_mask else None)
            return out, (self.get_seq_len(lens) if self.use_mask else lens)
        pad = self.conv.padding[0]
        eng = solo_engine(self, lambda: [UnitSpec(main=conv_spec(self.conv, None, pad, pad, PAD_ZERO, 'mconv'), src=0,
                                                   act=ACT_NONE, update_lens=self.use_mask)])
        out, lens_f = run_stack(eng, x, lens if self.use_mask else None, self.training)
        return out, (lens_f if self.use_mask else lens)


class GroupShuffle(nn.Module):
    """Channel shuffle between grouped convolutions (jasper.py:135-151): channel g*cpg + j moves to j*groups + g.  Not
    reachable from the config (``groups`` is never plumbed through ``_build_encoder``); plain torch view ops."""

    def __init__(self, groups, channels):
        super(GroupShuffle, self).__init__()
        self.groups = groups
        self.channels_per_group = channels // groups

    def forward(self, x):
        n, t = x.shape[0], x.shape[-1]
        grouped = x.reshape(n, self.groups, self.channels_per_group, t)
        return grouped.permute(0, 2, 1, 3).reshape(n, self.groups * self.channels_per_group, t)


class JasperBlock(nn.Module):
    __constants__ = ["conv_mask", "separable", "residual_mode", "res", "mconv"]

    def __init__(self, inplanes, planes, repeat=3, kernel_size=11, kernel_size_factor=1, stride=1, dilation=1,
                 padding='same', dropout=0, activation=None, residual=True, groups=1, separable=False, heads=-1,
                 normalization="batch", norm_groups=1, residual_mode='add', residual_panes=[], conv_mask=False):
        super(JasperBlock, self).__init__()
        if padding != "same":
            raise ValueError("currently only 'same' padding is supported")
        if normalization != "batch":
            if normalization in ("group", "instance", "layer"):
                raise NotImplementedError(f'normalization={normalization!r} is not reachable from the config '
                                          '(jasper.py:440-449) and is not built')
            raise ValueError(f"Normalization method ({normalization}) does not match one of [batch, layer, group, instance].")
        if groups != 1 or heads != -1 or len(residual_panes) or residual_mode != 'add':
            raise NotImplementedError('groups / heads / dense residual / max residual are not reachable from the config')
        if activation is not None and not isinstance(activation, nn.ReLU):
            raise NotImplementedError('Jasper._build_encoder always passes nn.ReLU() (jasper.py:448)')
        kernel_size_factor = float(kernel_size_factor)
        if type(kernel_size) in (list, tuple):
            kernel_size = [compute_new_kernel_size(k, kernel_size_factor) for k in kernel_size][0]
        else:
            kernel_size = compute_new_kernel_size(kernel_size, kernel_size_factor)
        padding_val = get_same_padding(kernel_size, stride, dilation)
        self.conv_mask = conv_mask
        self.separable = separable
        self.residual_mode = residual_mode
        self.repeat = repeat
        self.dropout = dropout
        self.kernel_size, self.stride, self.dilation, self.padding_val = kernel_size, stride, dilation, padding_val

        inplanes_loop = inplanes
        conv = nn.ModuleList()
        for _ in range(repeat - 1):
            conv.extend(self._get_conv_bn_layer(inplanes_loop, planes, kernel_size=kernel_size, stride=stride,
                                                dilation=dilation, padding=padding_val, separable=separable))
            conv.extend(self._get_act_dropout_layer(drop_prob=dropout))
            inplanes_loop = planes
        conv.extend(self._get_conv_bn_layer(inplanes_loop, planes, kernel_size=kernel_size, stride=stride,
                                            dilation=dilation, padding=padding_val, separable=separable))
        self.mconv = conv
        self.dense_residual = residual
        if residual:
            res_list = nn.ModuleList()
            res_list.append(nn.ModuleList(self._get_conv_bn_layer(inplanes, planes, kernel_size=1)))
            self.dense_residual = False
            self.res = res_list
        else:
            self.res = None
        self.mout = nn.Sequential(*self._get_act_dropout_layer(drop_prob=dropout))

    def _get_conv(self, in_channels, out_channels, kernel_size=11, stride=1, dilation=1, padding=0, bias=False,
                  groups=1):
        if not self.conv_mask:            # jasper.py:288-298: a bare nn.Conv1d -- no mask, no length update, key ``mconv.{j}.weight``
            return Conv1d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, padding=padding,
                          bias=bias, groups=groups, init='xavier_uniform')
        return MaskedConv1d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, padding=padding,
                            bias=bias, groups=groups, use_mask=True)

    def _get_conv_bn_layer(self, in_channels, out_channels, kernel_size=11, stride=1, dilation=1, padding=0,
                           separable=False):
        if separable:
            layers = [self._get_conv(in_channels, in_channels, kernel_size, stride=stride, dilation=dilation,
                                     padding=padding, groups=in_channels),
                      self._get_conv(in_channels, out_channels, kernel_size=1, stride=1, dilation=1, padding=0)]
        else:
            layers = [self._get_conv(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation,
                                     padding=padding)]
        layers.append(BatchNorm1d(out_channels, eps=1e-3, momentum=0.1))
        return layers

    def _get_act_dropout_layer(self, drop_prob=0.2):
        return [_NoParams('ReLU'), _NoParams(f'Dropout(p={drop_prob})')]

    def units(self, a_in: int, next_act: int, name: str, mask_last_output: bool) -> List[UnitSpec]:
        """engine units of this block; ``a_in`` = index of the block-input activation, ``next_act`` = index the
        first unit's output will get"""
        mods = list(self.mconv)
        groups = []                       # (depthwise MaskedConv1d or None, conv MaskedConv1d, BatchNorm1d)
        i = 0
        bare = lambda m: m.conv if isinstance(m, MaskedConv1d) else m          # noqa: E731 -- the Conv1d parameter holder
        while i < len(mods):
            if isinstance(mods[i], (MaskedConv1d, Conv1d)):
                if self.separable:
                    groups.append((bare(mods[i]), bare(mods[i + 1]), mods[i + 2]))
                    i += 3
                else:
                    groups.append((None, bare(mods[i]), mods[i + 1]))
                    i += 2
            else:
                i += 1
        out = []
        src = a_in
        for r, (dwm, mc, bn) in enumerate(groups):
            last = r == len(groups) - 1
            if dwm is None:
                spec = conv_spec(mc, bn, self.padding_val, self.padding_val, PAD_ZERO, f'{name}.mconv{r}')
                dws = None
            else:
                dws = conv_spec(dwm, None, self.padding_val, self.padding_val, PAD_ZERO, f'{name}.dw{r}', depthwise=True)
                spec = conv_spec(mc, bn, 0, 0, PAD_ZERO, f'{name}.pw{r}')
            # an activation is masked where it is WRITTEN, for the masked_fill of the MaskedConv1d that READS it
            # (jasper.py:116-119): inside the block that is this block's own conv_mask, behind its last unit the next block's
            u = UnitSpec(main=spec, src=src, act=ACT_RELU, drop_p=float(self.dropout), update_lens=self.conv_mask,
                         mask_out=mask_last_output if last else self.conv_mask, dw=dws)
            if last and self.res is not None:
                rc, rbn = self.res[0][0], self.res[0][1]
                u.res = conv_spec(bare(rc), rbn, 0, 0, PAD_ZERO, f'{name}.res')
                u.res_src = a_in
            out.append(u)
            src = next_act + r
        return out

    def forward(self, input_):
        """((x [N, C, T], lens)) -> (block output [N, planes, T'], lens') -- jasper.py:379-419 -- as an open engine of the
        block's units with autograd.  The output is not length-masked (the NEXT MaskedConv1d does that, jasper.py:116-119)."""
        xs, lens = (input_[0], input_[1]) if len(input_) == 2 else (input_[0], None)
        if isinstance(xs, (list, tuple)):
            xs = xs[-1]
        eng = solo_engine(self, lambda: self.units(0, 1, 'block', mask_last_output=False))
        if getattr(self, '_debug_keep_ctx', False):        # test hook: expose the engine's saved activations
            out, lens_f, self._last_ctx = run_stack(eng, xs, lens, self.training, keep_ctx=True)
            return out, lens_f
        out, lens_f = run_stack(eng, xs, lens, self.training)
        return out, lens_f


class Jasper(ConvCTCASR):
    def __init__(self, cfg):
        super(Jasper, self).__init__(cfg)
        self.mid_layers = cfg.mid_layers
        self.input_size = feature_size(cfg, self.audio_conf)
        self.precision = default_precision(cfg)
        self.check_nan = True                    # jasper.py:474 asserts on every forward (host sync)
        width = self._build_encoder(cfg)
        # classifier: a plain (unmasked) 1x1 Conv1d WITH bias, inside a Sequential (key ``final_layer.0``), jasper.py:432-434
        self.final_layer = nn.Sequential(Conv1d(width, len(self.labels), kernel_size=1, stride=1, init='xavier_uniform'))
        self.final_layer.apply(init_weights)

    def _build_encoder(self, cfg) -> int:
        """the first ``mid_layers`` rows of cfg.jasper_blocks as a Sequential of JasperBlock (jasper.py:436-453); every block
        gets its own ReLU instance request (the engine fuses the activation anyway); returns the encoder's output width"""
        widths = [self.input_size]
        blocks = []
        for row in cfg.jasper_blocks[: cfg.mid_layers]:
            kw = block_kwargs(row)
            blocks.append(JasperBlock(inplanes=widths[-1], activation=nn.ReLU(), **kw))
            widths.append(kw['planes'])
        self.jasper_encoder = nn.Sequential(*blocks)
        self.jasper_encoder.apply(init_weights)
        return widths[-1]

    @property
    def scaling_factor(self):
        """input frames per output frame: the stride of each block's FIRST conv, multiplied (jasper.py:455-459); once"""
        cached = self.__dict__.get('_scaling_factor')
        if cached is None:
            cached = self.__dict__['_scaling_factor'] = math.prod(b.mconv[0].conv.stride[0] for b in self.jasper_encoder)
        return cached

    def _build_engine(self) -> StackEngine:
        units: List[UnitSpec] = []
        a_in = 0
        blocks = list(self.jasper_encoder)
        for b, blk in enumerate(blocks):
            # the classifier is a plain nn.Conv1d (jasper.py:433,468): the last block's output is NOT masked, and neither
            # is the input of a block of plain convolutions (conv_mask: False)
            us = blk.units(a_in, len(units) + 1, f'block{b}', mask_last_output=b != len(blocks) - 1 and blocks[b + 1].conv_mask)
            units += us
            a_in = len(units)
        head = conv_spec(self.final_layer[0], None, 0, 0, PAD_ZERO, 'head')
        return StackEngine(units, head, len(self.labels), precise=self.precision == 'fp32', fp8=self.precision == 'fp8')

    def engine(self) -> StackEngine:
        return self._cached_engine(self._build_engine)

    def forward(self, xs, input_lengths):
        """[batch, channels, time], lengths -> ([batch, time', labels], lengths) (jasper.py:462-475):
        log_softmax in training, softmax in eval (reference quirk, kept)."""
        mode = 0 if self.training else 1
        if getattr(self, '_debug_keep_ctx', False):
            out, lens_f, self._last_ctx = run_stack(self.engine(), xs, input_lengths, self.training, softmax_mode=mode,
                                                    keep_ctx=True)
        else:
            out, lens_f = run_stack(self.engine(), xs, input_lengths, self.training, softmax_mode=mode)
        output_lengths = lens_f.to(dtype=torch.int64).cpu() if lens_f is not None else None
        if self.check_nan:
            assert not bool(torch.isnan(out).any())          # is there any NaN in the result?
        return out, output_lengths
