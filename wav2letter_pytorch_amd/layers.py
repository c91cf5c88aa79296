"""Parameter holders with the reference's state-dict names, and the autograd bridge
between a module tree and the StackEngine."""
from __future__ import annotations

import math
import os
from itertools import chain
from typing import Callable, Optional

import torch
import torch.nn as nn

from .engine import ACT_NONE, PAD_ZERO, ConvSpec, StackEngine, UnitSpec


def default_precision(cfg=None) -> str:
    """'bf16' (bf16 operands, fp32 accumulate: the production mode), 'fp32' (split-bf16, three MFMA passes per
    product: the parity mode against the fp32 reference) or 'fp8' (forward convolutions on e4m3 operands through the
    block-scaled MFMA at twice the bf16 rate; gradients, statistics and the classifier as in bf16 mode)."""
    p = None
    if cfg is not None:
        try:
            p = cfg.get('precision', None)
        except Exception:
            p = None
    p = p or os.environ.get('W2L_PRECISION', 'bf16')
    if p not in ('bf16', 'fp32', 'fp8'):
        raise ValueError(f"precision must be 'bf16', 'fp32' or 'fp8', got {p!r}")
    return p


class Conv1d(nn.Module):
    """Holds nn.Conv1d's parameters (wav2letter.py:35-36, jasper.py:96-105).  ``weight`` has the
    logical shape [out, in/groups, k] of nn.Conv1d but is stored physically as [k, out, in]
    (tap-major, channels contiguous): that is the layout the wgrad kernel writes and the
    pack kernel reads coalesced.  state_dict() / load_state_dict() see the logical shape."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 init='torch_default'):
        super().__init__()
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (k,)
        self.stride = (stride,)
        self.padding = (padding,)
        self.dilation = (dilation,)
        self.groups = groups
        phys = torch.empty(k, out_channels, in_channels // groups)
        self.weight = nn.Parameter(phys.permute(1, 2, 0))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters(init)

    def reset_parameters(self, init='torch_default'):
        # draw into a contiguous tensor so a given seed reproduces nn.Conv1d's values exactly
        tmp = torch.empty(self.weight.shape)
        if init == 'xavier_uniform':                      # jasper.py:29-36
            nn.init.xavier_uniform_(tmp, gain=1.0)
        else:                                             # nn.Conv1d default (wav2letter.py:35-36)
            nn.init.kaiming_uniform_(tmp, a=math.sqrt(5))
        with torch.no_grad():
            self.weight.copy_(tmp)
            if self.bias is not None:
                fan_in = self.weight.shape[1] * self.weight.shape[2]
                bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                self.bias.copy_(torch.empty(self.bias.shape).uniform_(-bound, bound))

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, '
                f'padding={self.padding}, dilation={self.dilation}, bias={self.bias is not None}')

    def forward(self, x):
        """[N, C_in, T] -> [N, C_out, T'] fp32: the convolution alone (zero padding ``self.padding``), as a one-unit open
        engine with autograd.  Inside a model the parent's engine runs it fused with its neighbours instead."""
        if self.groups != 1:
            return depthwise_forward(self, x, None)        # groups == channels; other group counts raise there
        pad = self.padding[0]
        eng = solo_engine(self, lambda: [UnitSpec(main=conv_spec(self, None, pad, pad, PAD_ZERO, 'conv'), src=0, act=ACT_NONE)])
        out, _ = run_stack(eng, x, None, self.training)
        return out


class BatchNorm1d(nn.Module):
    """Holds nn.BatchNorm1d's parameters and buffers (wav2letter.py:37, jasper.py:363)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine = True
        self.track_running_stats = True
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f'{self.num_features}, eps={self.eps}, momentum={self.momentum}'

    def forward(self, x):
        raise RuntimeError('BatchNorm1d holds parameters and running statistics only: the kernels normalise the output of '
                           'the convolution in front of it (Conv1dBlock / JasperBlock), there is no conv-less BatchNorm pass')


class _DepthwiseFn(torch.autograd.Function):
    """A depthwise convolution (``nn.Conv1d(C, C, k, groups=C)``, jasper.py:96-105,319-330) called on its own: the kernels
    of csrc/dwconv.hip through the step engine's own depthwise helpers, with autograd.  Inside a separable JasperBlock the
    model's engine runs the same kernels in front of the pointwise convolution instead."""

    @staticmethod
    def forward(ctx, x, conv, lens, weight, bias):
        import ctypes as C
        from . import _lib
        from ._lib import check, lib, ptr, stream_ptr
        from .engine import Act, padded_channels
        _lib.require_device(x)
        prec = getattr(conv, 'precision', None) or default_precision()
        eng = StackEngine([], None, 0, precise=prec == 'fp32')
        pad = conv.padding[0]
        spec = conv_spec(conv, None, pad, pad, PAD_ZERO, 'depthwise', depthwise=True)
        x = x.contiguous().float()
        n, c, t = x.shape
        cp = padded_channels(c)
        dev = x.device
        lens_dev = None if lens is None else lens.to(device=dev, dtype=torch.int32)
        a_hi = torch.empty(n, t + 2 * pad, cp, dtype=torch.bfloat16, device=dev)
        a_lo = torch.empty_like(a_hi) if eng.precise else None
        check(lib.w2l_nct_to_ntc(ptr(x), n, c, t, cp, pad, pad, PAD_ZERO, ptr(lens_dev), ptr(a_hi), ptr(a_lo), stream_ptr()),
              'w2l_nct_to_ntc')
        src = Act(a_hi, a_lo, n, t, c, cp, pad, pad, PAD_ZERO, lens_dev)
        mid = eng._dw_forward(spec, src, None)              # the output is not masked: the NEXT MaskedConv1d does that
        out = mid.hi[:, :, :c].float()
        if mid.lo is not None:
            out = out + mid.lo[:, :, :c].float()
        out = out.transpose(1, 2).contiguous()
        if bias is not None:
            out = out + bias.detach()[None, :, None]
        ctx.state = (eng, spec, src, mid, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        eng, spec, src, mid, has_bias = ctx.state
        n, c, t, cp = src.N, src.C, src.T, src.CP
        gp = torch.zeros(n, mid.T, cp, dtype=torch.float32, device=g.device)
        gp[:, :, :c] = g.float().transpose(1, 2)
        grads = {}
        gsrc = eng._dw_backward(spec, (gp, 0, 0, PAD_ZERO, mid.T), src, mid, True, grads)
        dxp, pl, _, _, per = gsrc[:5]
        dx = dxp.view(n, per, cp)[:, pl:pl + t, :c].float()
        if src.lens is not None:                             # masked_fill on the input (jasper.py:116-119)
            dx = dx * (torch.arange(t, device=dx.device)[None, :, None] < src.lens.long()[:, None, None])
        ctx.state = None
        return dx.transpose(1, 2).contiguous(), None, None, grads[id(spec.weight)], (g.sum((0, 2)) if has_bias else None)


def depthwise_forward(conv: 'Conv1d', x, lens):
    """[N, C, T] -> [N, C, T'] through the depthwise kernels; ``lens`` (optional) zeroes input frames t >= len first"""
    if conv.groups != conv.in_channels or conv.in_channels != conv.out_channels:
        raise NotImplementedError('grouped convolution with 1 < groups < channels is not built (not reachable from the '
                                  'config: jasper.py:440-449 never passes groups)')
    return _DepthwiseFn.apply(x, conv, lens, conv.weight, conv.bias)


def conv_spec(conv: Conv1d, bn: Optional[BatchNorm1d], pad_l: int, pad_r: int, pad_mode: int, name: str = '',
              depthwise: bool = False) -> ConvSpec:
    if depthwise:
        if conv.groups != conv.in_channels or conv.in_channels != conv.out_channels:
            raise NotImplementedError('only depthwise (groups == channels) grouped convolutions are built')
    elif conv.groups != 1:
        raise NotImplementedError('grouped convolution (1 < groups < channels) is not reachable from the config')
    spec = ConvSpec(weight=conv.weight, bias=conv.bias, kernel=conv.kernel_size[0], stride=conv.stride[0],
                    dilation=conv.dilation[0], pad_l=pad_l, pad_r=pad_r, pad_mode=pad_mode, name=name)
    if bn is not None:
        spec.bn_weight, spec.bn_bias = bn.weight, bn.bias
        spec.running_mean, spec.running_var = bn.running_mean, bn.running_var
        spec.num_batches_tracked = bn.num_batches_tracked
        spec.eps, spec.momentum = bn.eps, bn.momentum
    return spec


class _StackFn(torch.autograd.Function):
    """One autograd node for the whole conv stack + classifier + (log_)softmax."""

    @staticmethod
    def forward(ctx, x, engine: StackEngine, lens, training: bool, softmax_mode: int, holder: dict, *params):
        want_dx = bool(x.requires_grad)
        out, ectx = engine.forward(x, lens, training, softmax_mode, want_input_grad=want_dx)
        ctx.engine, ctx.ectx = engine, ectx
        if holder.get('keep_ctx'):
            holder['ctx'] = ectx
        holder['lens_out'] = ectx['lens_out']
        return out

    @staticmethod
    def backward(ctx, g):
        from . import replay
        rp = ctx.engine.__dict__.get('_replayer')
        if rp is not None:                 # an eager backward pass: whatever it holds back is the eager engine's to launch
            rp.before_eager()
            rp.pending = None
        replay._last_backward[0] = None
        grads = ctx.engine.backward(ctx.ectx, g)
        dx = ctx.ectx.get('input_grad')
        ctx.ectx = None
        return (dx, None, None, None, None, None, *grads)


def run_stack(engine: StackEngine, x, lens, training: bool, softmax_mode: int = 0, keep_ctx: bool = False):
    """Returns (out, lens_out[, engine ctx when keep_ctx: test hook that exposes the saved activations]).
    A warm training step is recorded once and replayed from then on (replay.py: one w2l_replay call per phase)."""
    from . import replay
    holder = {'keep_ctx': keep_ctx}
    rp = replay.replayer_for(engine) if x.is_cuda else None
    if rp is not None:
        plan = rp.plan_forward(x, lens, training, softmax_mode, keep_ctx)
        if plan is not None:
            out = replay._ReplayFn.apply(x, engine, lens, softmax_mode, holder, plan[0], plan[1], *engine.parameters())
            return out, holder.get('lens_out')
        rp.before_eager()
        rp.flush_pending()                 # (a recorded set's held-back gradients, replayed; anything else is left to forward())
    out = _StackFn.apply(x, engine, lens, training, softmax_mode, holder, *engine.parameters())
    if keep_ctx:
        return out, holder.get('lens_out'), holder.get('ctx')
    return out, holder.get('lens_out')


def solo_engine(owner: nn.Module, units_fn: Callable[[], list]) -> StackEngine:
    """Open (head-less) engine for a module called on its own -- Conv1dBlock, MaskedConv1d, JasperBlock, Conv1d --
    cached on the module and rebuilt when a parameter / buffer object, its device or the precision changes."""
    prec = getattr(owner, 'precision', None) or default_precision()
    key = tuple((id(t), t.device) for t in chain(owner.parameters(), owner.buffers())) + (prec,)
    hit = owner.__dict__.get('_solo_engine')
    if hit is None or hit[0] != key:
        from .base_asr_models import EngineSlot        # a copy / pickle of the module gets an empty slot, not this engine
        hit = EngineSlot((key, StackEngine(units_fn(), None, 0, precise=prec == 'fp32', fp8=prec == 'fp8')))
        owner.__dict__['_solo_engine'] = hit
    return hit[1]
