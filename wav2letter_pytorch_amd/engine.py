"""Step engine: runs the conv stack of Wav2Letter / Jasper forward and backward on the
HIP kernels of libw2l_hip.so, one explicit pass each -- no autograd graph inside.

The stack is a list of *units*; a unit is what the reference expresses as
  Conv1dBlock (wav2letter.py:40-47): reflect-pad -> conv -> BN -> dropout -> clamp, or
  one conv-BN(-act-dropout) step of a JasperBlock (jasper.py:379-419), the last step of a
  block carrying the residual branch (1x1 conv + BN of the block input, add, ReLU).
Activations between units live in HBM as channels-last bf16 buffers that are already
padded for their consumer (reflect or zero halo), so the implicit-GEMM conv kernel does
pure address arithmetic.  The backward pass walks the units in reverse, producing for
each unit: BN/activation backward -> dy (zero-haloed) -> wgrad -> dgrad, and hands every
weight gradient to an optional callback as soon as it exists (data-parallel all-reduce
overlap, see distributed.py).
"""
from __future__ import annotations

import atexit
import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from . import wgrad_groups as WG
from ._lib import BnActDesc, GradSrc, check, lib, ptr, stream_ptr

ACT_NONE, ACT_CLAMP20, ACT_RELU = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1


def _none():
    return None


class _Volatile(dict):
    """per-Parameter device state kept in ``Parameter.__dict__`` (operand packs, e4m3 copies, events, pinned buffers): a
    pickled or deep-copied Parameter gets None in its place and rebuilds the state on first use"""

    def __reduce__(self):
        return (_none, ())

    def __deepcopy__(self, memo):
        return None


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def zeros(shape, dtype, device) -> torch.Tensor:
    """torch.zeros through the library's fill (w2l_fill_zero: an entry point, so a recorded launch list replays it)"""
    t = torch.empty(shape, dtype=dtype, device=device)
    if t.is_cuda:
        check(lib.w2l_fill_zero(ptr(t), t.numel() * t.element_size(), stream_ptr()), 'w2l_fill_zero')
    else:
        t.zero_()
    return t


def zero_(t: torch.Tensor) -> torch.Tensor:
    if t.is_cuda and t.is_contiguous():
        check(lib.w2l_fill_zero(ptr(t), t.numel() * t.element_size(), stream_ptr()), 'w2l_fill_zero')
    else:
        t.zero_()
    return t


def _py(fn, *args):
    """a Python callback between launches (optimizer / reducer hooks): run now, and -- while a phase is being recorded -- kept
    in sequence in the record, with the stream that is current here"""
    rec = _lib.recording()
    if rec is None:
        return fn(*args)
    rec.python(fn, *args, stream=torch.cuda.current_stream())


def padded_channels(c: int) -> int:
    return roundup(c, 64)


# --------------------------------------------------------------------------- specs
@dataclass
class ConvSpec:
    """One nn.Conv1d (+ optional BatchNorm1d) of the stack; tensors are the module's own
    Parameters / buffers (state-dict names stay the reference's)."""
    weight: torch.Tensor                  # logical [Cout, Cin, Kw]; physical [Kw, Cout, Cin] preferred
    bias: Optional[torch.Tensor]
    kernel: int
    stride: int
    dilation: int
    pad_l: int
    pad_r: int
    pad_mode: int
    bn_weight: Optional[torch.Tensor] = None
    bn_bias: Optional[torch.Tensor] = None
    running_mean: Optional[torch.Tensor] = None
    running_var: Optional[torch.Tensor] = None
    num_batches_tracked: Optional[torch.Tensor] = None
    eps: float = 1e-3
    momentum: float = 0.1
    name: str = ''

    @property
    def has_bn(self) -> bool:
        return self.bn_weight is not None

    @property
    def cout(self) -> int:
        return self.weight.shape[0]

    @property
    def cin(self) -> int:
        return self.weight.shape[1]

    def params(self) -> List[torch.Tensor]:
        out = [self.weight]
        if self.bias is not None:
            out.append(self.bias)
        if self.has_bn:
            out += [self.bn_weight, self.bn_bias]
        return out


@dataclass
class UnitSpec:
    main: ConvSpec
    src: int                              # index of the activation the main conv reads
    act: int = ACT_NONE
    drop_p: float = 0.0
    dw: Optional[ConvSpec] = None         # depthwise conv in front of ``main`` (then a 1x1 pointwise conv): the
                                          # separable form of jasper.py:318-341; ``src`` feeds the depthwise conv
    res: Optional[ConvSpec] = None        # residual branch conv (+BN) -- jasper.py:241-255,400-410
    res_src: Optional[int] = None
    update_lens: bool = False             # the conv is a MaskedConv1d: lens <- get_seq_len(lens) (jasper.py:109-121)
    mask_out: bool = False                # zero frames t >= len of the output: the NEXT MaskedConv1d's masked_fill
                                          # (jasper.py:116-119), applied where the activation is written


@dataclass
class Act:
    """A padded channels-last activation buffer [N][pad_l + T + pad_r][CP]."""
    hi: torch.Tensor
    lo: Optional[torch.Tensor]
    N: int
    T: int
    C: int
    CP: int
    pad_l: int
    pad_r: int
    pad_mode: int
    lens: Optional[torch.Tensor] = None   # int32 [N] on device: frames t >= lens[n] are zero
    q: Optional[torch.Tensor] = None      # fp8 mode: the same buffer as OCP e4m3 bytes, value * q_scale
    q_scale: float = 1.0

    @property
    def rows(self) -> int:
        return self.pad_l + self.T + self.pad_r


@dataclass
class _PackedW:
    version: int
    fwd_hi: torch.Tensor
    fwd_lo: Optional[torch.Tensor]
    dgr_hi: torch.Tensor
    dgr_lo: Optional[torch.Tensor]
    cinp: int
    coutp: int
    src_ptr: int = 0
    ready: Optional['_lib.Event'] = None            # set by optim.FusedSGD when the pack was written on its side stream: the
                                                    # weight's OWN event (weight_event), recorded again by every update


@dataclass
class _UnitCtx:
    unit: UnitSpec
    y: torch.Tensor = None
    y2: Optional[torch.Tensor] = None
    Tout: int = 0
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    mean: Optional[torch.Tensor] = None
    invstd: Optional[torch.Tensor] = None
    scale2: Optional[torch.Tensor] = None
    shift2: Optional[torch.Tensor] = None
    mean2: Optional[torch.Tensor] = None
    invstd2: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None
    seed: int = 0
    offset: int = 0
    out_index: int = 0
    lens_out: Optional[torch.Tensor] = None
    mid: Optional['Act'] = None           # depthwise output = pointwise input (separable units)
    keep: list = field(default_factory=list)


_dropout_calls = 0
# measure-and-pick of the conv block shape per problem shape (synchronises once per new shape); W2L_AUTOTUNE=0 keeps
# the library's cost model
AUTOTUNE = os.environ.get('W2L_AUTOTUNE', '1') != '0'
_tuned_shapes = set()
TUNE_REPS = int(os.environ.get('W2L_TUNE_REPS', '2'))      # timed launches per candidate configuration
# W2L_TUNE_CACHE=<file>: measured choices are loaded at import and written back (atomically) after a step that
# measured new shapes, so later processes (and the other DP ranks) skip the measuring launches.
TUNE_CACHE = os.environ.get('W2L_TUNE_CACHE') or None
_tune_state = {'dirty': False}
_wgroup_plans = {}                 # (convolutions of a backward pass, W2L_WGRAD_GROUPS) -> groups (wgrad_groups.plan)
_wgroup_forms = {}                 # group signature -> measured block form of its launch, -1: one by one


def load_tune_cache(path: str) -> int:
    n = lib.w2l_tune_load(path.encode())
    if n < 0:
        check(1, 'w2l_tune_load')
    # the step engine's own measured choices ride in the same file (lines the library skips): the block form of every
    # grouped weight-gradient launch, -1 = its members one by one (StackEngine._wgrad_group_measure)
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) == 6 and t[0] == 'wgroup':
                try:
                    layers = tuple(tuple(int(v) for v in l.split(',')) for l in t[5].split(';'))
                    _wgroup_forms[(layers, int(t[1]), int(t[2]), int(t[3]))] = int(t[4])
                    n += 1
                except ValueError:
                    pass
    return n


def save_tune_cache(path: str):
    tmp = '%s.tmp.%d' % (path, os.getpid())
    check(lib.w2l_tune_save(tmp.encode()), 'w2l_tune_save')
    with open(tmp, 'a') as f:
        for (layers, n, tout, dil), form in sorted(_wgroup_forms.items()):
            f.write('wgroup %d %d %d %d %s\n' % (n, tout, dil, form, ';'.join(','.join(str(v) for v in l) for l in layers)))
    os.replace(tmp, path)


def _flush_tune_cache():
    if _tune_state['dirty'] and TUNE_CACHE:
        save_tune_cache(TUNE_CACHE)
    _tune_state['dirty'] = False


if TUNE_CACHE and os.path.exists(TUNE_CACHE):
    load_tune_cache(TUNE_CACHE)
atexit.register(_flush_tune_cache)

# callables run at the end of every StackEngine.forward (optim.FusedSGD releases the previous step's gradients there, once
# the forward has waited for every update that read them)
AFTER_FORWARD: list = []

# optional kernel timer (bench.py): list of (kernel_name, flops, start_event, end_event)
KERNEL_TIMER: Optional[list] = None


class _timed:
    def __init__(self, name, flops):
        self.name, self.flops = name, flops

    def __enter__(self):
        if KERNEL_TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if KERNEL_TIMER is not None:
            self.e.record()
            KERNEL_TIMER.append((self.name, self.flops, self.s, self.e))


def _phys_strides(w: torch.Tensor):
    """element strides (s_co, s_ci, s_kw) of the logical [Cout,Cin,Kw] weight"""
    return w.stride(0), w.stride(1), w.stride(2)


def pack_weights(conv: ConvSpec, precise: bool, need_dgrad: bool = True) -> _PackedW:
    """fp32 master weights -> bf16 GEMM operand layouts (cached per parameter version)."""
    w = conv.weight
    cache = getattr(w, '_w2l_pack', None)          # lives and dies with the Parameter object
    if cache is None:
        cache = _Volatile()
        w._w2l_pack = cache
    hit = cache.get(precise)
    recording = _lib.recording() is not None
    if hit is not None and hit.fwd_hi.device == w.device and hit.src_ptr == w.data_ptr():
        fused = w.__dict__.get('_w2l_ready_ev')
        if hit.version == w._version:
            if hit.ready is not None:      # the optimizer updated + packed this weight on its own stream: order after it
                hit.ready.wait()
                hit.ready = None
            elif recording and fused is not None and _in_forward[0]:
                # a recorded forward pass must not depend on which flags happened to be up when it was recorded (an evaluation
                # forward in front of it has consumed them): it ALWAYS waits for the weight's event -- free when it has fired
                fused.wait()
            if not recording or fused is not None:
                return hit
        # the weight moved since it was packed (an optimizer that does not pack: torch.optim.*, Novograd on a layer it does
        # not fuse) -- or a forward pass is being recorded and no fused optimizer maintains this pack, so that the replayed
        # pass must repack every step whatever the cache says now: packed again INTO THE SAME BUFFERS (every reader of the
        # old contents is ahead of this launch on the caller's stream; the weight-gradient stream never reads operand packs)
        cout, cin, kw = w.shape
        s_co, s_ci, s_kw = _phys_strides(w)
        check(lib.w2l_pack_weights(ptr(w), s_co, s_ci, s_kw, cout, cin, kw, hit.coutp, hit.cinp, ptr(hit.fwd_hi), ptr(hit.fwd_lo),
                                   ptr(hit.dgr_hi), ptr(hit.dgr_lo), stream_ptr()), 'w2l_pack_weights')
        hit.version, hit.ready = w._version, None
        return hit
    cout, cin, kw = w.shape
    coutp, cinp = padded_channels(cout), padded_channels(cin)
    dev = w.device
    if recording:
        _lib.poison('operand pack buffers created')        # (new buffers every step: not a step to record; the next one hits)
    fwd_hi = torch.empty(kw, coutp, cinp, dtype=torch.bfloat16, device=dev)
    dgr_hi = torch.empty(kw, cinp, coutp, dtype=torch.bfloat16, device=dev)
    fwd_lo = torch.empty_like(fwd_hi) if precise else None
    dgr_lo = torch.empty_like(dgr_hi) if precise else None
    s_co, s_ci, s_kw = _phys_strides(w)
    check(lib.w2l_pack_weights(ptr(w), s_co, s_ci, s_kw, cout, cin, kw, coutp, cinp, ptr(fwd_hi), ptr(fwd_lo),
                               ptr(dgr_hi), ptr(dgr_lo), stream_ptr()), 'w2l_pack_weights')
    pk = _PackedW(w._version, fwd_hi, fwd_lo, dgr_hi, dgr_lo, cinp, coutp, w.data_ptr())
    cache[precise] = pk
    return pk


def weight_event(w) -> '_lib.Event':
    """the ONE event of a conv weight that every fused update of it records (on whichever stream runs the update) and the next
    forward convolution of that layer waits for.  One persistent event per weight rather than a fresh one per step: a recorded
    forward pass waits for the same handles whichever step -- eager or replayed -- made the update before it."""
    ev = w.__dict__.get('_w2l_ready_ev')
    if ev is None:
        ev = w.__dict__['_w2l_ready_ev'] = _lib.Event()
        _wev_epoch[0] += 1                 # forward passes recorded before this weight had an event do not wait for it: stale
    return ev


_wev_epoch = [0]
_in_forward = [False]              # (the backward pass looks packs up too: it has nothing to wait for -- its forward already did)


def invalidate_packed(params) -> int:
    """Drop the cached bf16 operand packs of ``params`` (an iterable of Parameters, or a Module).

    The cache is keyed on ``Parameter._version``, which in-place writes through ``.data`` / raw pointers do not bump
    (old-style optimizers doing ``p.data.add_``, EMA, weight clipping, ``dist.broadcast(p.data)``).  Whoever updates
    weights that way calls this afterwards; the next forward then repacks from the fp32 master weights.  Weights updated
    through version-bumping in-place ops (torch.optim, optim.FusedSGD, novograd.Novograd) need nothing."""
    if isinstance(params, torch.nn.Module):
        params = params.parameters()
    n = 0
    for p in params:
        cache = getattr(p, '_w2l_pack', None)
        if cache:
            cache.clear()
            n += 1
    return n


def _padded_vec(v: Optional[torch.Tensor], cp: int, fill: float) -> Optional[torch.Tensor]:
    """per-channel fp32 vector padded to CP (no copy when already that size)"""
    if v is None:
        return None
    v = v.detach()
    if v.dtype != torch.float32:
        _lib.poison('a per-channel vector that is not fp32')
        v = v.float()
    if v.numel() == cp and v.is_contiguous():
        return v
    if v.is_cuda and v.is_contiguous():
        out = torch.empty(cp, dtype=torch.float32, device=v.device)
        check(lib.w2l_pad_vec_f32(ptr(v), v.numel(), ptr(out), cp, float(fill), stream_ptr()), 'w2l_pad_vec_f32')
        return out
    out = torch.full((cp,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v
    return out


SPLITK_WS_CAP = 1 << 30            # bytes; larger problems simply lose the split configurations that do not fit
_splitk_ws = {}                    # device index -> zero-initialised workspace (tickets + fp32 slabs) of the main stream


def _splitk_workspace(dev, n, cout, tout):
    """The split-K workspace of ``dev`` (include/w2l_hip.h: w2l_conv1d_igemm_ws), grown on demand.  Every implicit-GEMM launch
    of the engine is on the caller's stream, so one workspace per device serves them all."""
    need = min(int(lib.w2l_conv_splitk_workspace_bytes(n, cout, tout)), SPLITK_WS_CAP)
    ws = _splitk_ws.get(dev.index)
    if ws is None or ws.numel() < need:
        _lib.poison('split-K workspace grown')
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        _splitk_ws[dev.index] = ws
    return ws


_side_streams = {}                 # device index -> the stream weight gradients run on


def _side_stream(dev, main=None) -> 'torch.cuda.Stream':
    # a plain default-priority stream.  Measured and rejected: hipStreamCreateWithPriority (lowest or highest: 18.8 / 20.1
    # instead of 14.3 ms per step) and hipExtStreamCreateWithCUMask with 7/8, 6/8 or 5/8 of the CUs (16.6-16.7 instead of
    # 13.7 ms per step, round 2): any special queue makes the weight gradients crawl.
    # WHICH pool stream: the first one measured to run beside the main stream (streams.py) -- a stream that shares the main
    # stream's hardware queue / pipe serialises the step (20.0 instead of 14.0 ms), and which one does depends on every
    # stream created in the process before it (RCCL's included).
    st = _side_streams.get(dev.index)
    if st is None:
        from .streams import concurrent_stream
        _lib.poison('side stream created')
        st = concurrent_stream(dev, 'wgrad', main=main)
        _side_streams[dev.index] = st
    return st


# (the largest group: measured in the step -- tools/step_ab.py, profiles/r05_step_ab.txt -- groups of up to 3 layers are worth -0.16 ms
# on the headline step and -0.18 on Jasper 10x5, groups of 8 the same on the headline but +0.13 at N = 16: long-lived
# full-chip launches keep the data gradients of the main stream waiting)
WGROUP_MAX = int(os.environ.get('W2L_WGRAD_GROUP_MAX', '3'))
_wgrad_ws = {}                     # device index -> workspace of the stream the weight gradients run on
_retired_ws = []                   # outgrown workspaces (grown only while shapes are new, i.e. a handful of times)
# W2L_DETERMINISTIC=1: split weight-gradient reductions go through slabs summed in a fixed order instead of fp32 atomics
# (bit-reproducible gradients, no zero fills) -- measured 6 % slower on the Wav2Letter table (943 vs 1004 TFLOP/s), so opt-in
DETERMINISTIC_WGRAD = os.environ.get('W2L_DETERMINISTIC', '0') == '1'
lib.w2l_wgrad_deterministic(int(DETERMINISTIC_WGRAD))      # (a default-mode plan cache must not bring the atomics back)
# W2L_DEALT_WGRAD=0: no workspace for the weight gradients in the default mode, i.e. the dealt stream-K plans (include/w2l_hip.h)
# are neither measured nor run (round 4's plan space; the A/B switch of profiles/r05_step_ab.txt)
DEALT_WGRAD = os.environ.get('W2L_DEALT_WGRAD', '1') == '1'
DEFER_SPREAD = os.environ.get('W2L_DEFER_SPREAD', 'start')      # StackEngine._defer_positions
WGRAD_AFTER_DGRAD = os.environ.get('W2L_WGRAD_AFTER_DGRAD', '0') == '1'      # StackEngine._units_backward
# W2L_FUSED_BN_REDUCE: the BatchNorm-backward reduction of a layer formed in the epilogue of the data-gradient convolution
# that produces the gradient wrt its output (w2l_conv1d_dgrad_bnreduce_ws) instead of by w2l_bn_act_bwd_reduce: one kernel
# less per layer on the backward critical path.  '1' always, '0' never, 'auto' (default) for activations of fewer than
# FUSED_BN_REDUCE_MAX_ROWS frames (N x T').  Measured on one MI355X, same box per pair: Wav2Letter N=32 (16 000 frames)
# 13.75-13.85 ms per step with it, 13.71-13.94 without, four runs each (later pair 13.47 / 13.56) -- the epilogue costs the data
# gradients what the separate pass cost (conv_igemm_kernel 1166 instead of 1246 TFLOP/s) and a step of large kernels is
# bound by the sum of their work, not by the launches on its critical path; where the kernels are short it pays: Jasper 10x5
# N=16 (8 000 frames, 53 units) 19.39 / 19.78 with vs 19.98 / 20.17 without, Wav2Letter N=16 8.49 vs 8.57.
# Round 5: with the two-launch chain (FAST_BN_BWD) the separate passes win everywhere that was re-measured (tools/step_ab.py:
# Wav2Letter N = 8 / 16 / 32 -0.16 / -0.20 / -0.29 ms, Jasper 10x5 N = 16 -0.20 ms against the fused epilogue): default '0'.
FUSED_BN_REDUCE = os.environ.get('W2L_FUSED_BN_REDUCE', '0')
FUSED_BN_REDUCE_MAX_ROWS = 12288
# W2L_FOLD_BN_FINALIZE=1: the column sums of the BatchNorm-backward partials are re-formed by every block of the dy kernel
# for its own 64 channels (w2l_bn_act_bwd_apply_fin) instead of by a finalize launch of their own.  OFF by default: measured
# neutral to slightly slower (13.75-13.85 vs 13.71-13.80 ms; with the separate reduction pass 14.1 vs 13.7-13.9).
FOLD_BN_FINALIZE = os.environ.get('W2L_FOLD_BN_FINALIZE', '0') == '1'
# W2L_FOLD_BN_FWD (bf16 / fp8 training steps): the forward statistics finalize is folded into the BatchNorm-apply pass
# (w2l_bn_act_fwd_fin) -- the convolutions add their per-tile sums onto STAT_SLOTS rows (w2l_conv_stats_mode; fp32 atomics),
# every block of the apply pass re-reduces the rows of its 64 channels: one dependent launch less per layer on the forward's
# critical path.  Measured (tools/step_ab.py, interleaved in one process): the forward's BatchNorm gaps shrink from 48-52 to
# 41-45 us per layer; the headline step (16 000 frames per activation) by 0.00-0.04 ms -- inside the noise, and the forward
# stops being bit-reproducible (atomics) --, Wav2Letter N = 8 by 0.12 ms of 5.7, Jasper 10x5 N = 16 by 0.22 of 19.6 (106 forward
# launches fewer): 'auto' = on below FOLD_BN_FWD_MAX_ROWS frames, where the kernels are short and the launches count.
# 0 = round 4's two launches (bit-reproducible sums).
FOLD_BN_FWD = os.environ.get('W2L_FOLD_BN_FWD', 'auto')        # '1' always, '0' never, 'auto': for fewer than FOLD_BN_FWD_MAX_ROWS frames
FOLD_BN_FWD_MAX_ROWS = 12288
# W2L_FAST_BN_BWD (default 1): the BatchNorm-backward chain of a plain unit (bf16, one gradient source, no residual branch) in
# TWO launches instead of three (w2l_bn_act_bwd_reduce_slots + w2l_bn_act_bwd_apply_slots: sums onto STAT_SLOTS rows with fp32
# atomics, finalize folded into the dy pass, every load of a wave issued before the first use).  Measured -0.14 ms per step
# (13.14 -> 13.00, tools/step_ab.py, profiles/r05_step_ab.txt).  0 (or W2L_DETERMINISTIC=1) = round 4's chain.
FAST_BN_BWD = os.environ.get('W2L_FAST_BN_BWD', '1') == '1'      # (atomics: not bit-reproducible -- W2L_DETERMINISTIC=1 switches it off)
STAT_SLOTS = int(os.environ.get('W2L_STAT_SLOTS', '8'))


def _wgrad_workspace(dev, cin, cout, kw):
    """the workspace of the weight-gradient stream: sized for every slab plan in deterministic mode, otherwise for the dealt
    stream-K plans (the only ones that use it then: classic splits keep their atomics, plan order bit 6)"""
    if DETERMINISTIC_WGRAD:
        need = min(int(lib.w2l_wgrad_workspace_bytes(cin, cout, kw)), SPLITK_WS_CAP)
    else:
        need = min(int(lib.w2l_wgrad_dealt_workspace_bytes(cin, cout, kw)), SPLITK_WS_CAP)
        if need == 0:
            return None
    ws = _wgrad_ws.get(dev.index)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _retired_ws.append(ws)         # side-stream kernels may still be using it: never hand the memory back
        _lib.poison('weight-gradient workspace grown')
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        _wgrad_ws[dev.index] = ws
    return ws


def _igemm(x: Act, row_off: int, w_hi, w_lo, y, bias, stats, Cin, Cout, Tout, Kw, stride, dil, precise, alg_flops=0.0):
    """y = conv(x) through w2l_conv1d_igemm_ws; split-bf16 (3 launches, fp32 accumulate) when precise."""
    n = x.N
    bstride = x.rows * x.CP
    rows_total = n * x.rows - row_off
    esz = 2

    def xptr(t):
        return C.c_void_p(t.data_ptr() + row_off * x.CP * esz)

    st = stream_ptr()
    if not precise:
        ws = _splitk_workspace(x.hi.device, n, Cout, Tout)
        if AUTOTUNE:
            key = (n, Cin, Cout, Tout, Kw, stride, dil, stats is not None, x.hi.device.index)
            if key not in _tuned_shapes:       # once per shape and device, during the first (warm-up) step
                _tuned_shapes.add(key)
                _tune_state['dirty'] = True
                _lib.poison('measuring launches')
                check(lib.w2l_conv1d_igemm_tune_ws(xptr(x.hi), bstride, rows_total, ptr(w_hi), ptr(y),
                                                   int(y.dtype == torch.float32), ptr(bias), ptr(stats), n, Cin, Cout, Tout,
                                                   Kw, stride, dil, TUNE_REPS, ptr(ws), ws.numel(), st), 'w2l_conv1d_igemm_tune_ws')
                if stats is not None and stats.shape[0] <= 64:      # w2l_conv_stats_mode: the measuring launches ADDED to the rows
                    stats.zero_()
        with _timed('conv_igemm_kernel', alg_flops):
            check(lib.w2l_conv1d_igemm_ws(xptr(x.hi), bstride, rows_total, ptr(w_hi), ptr(y), int(y.dtype == torch.float32),
                                          0, ptr(bias), ptr(stats), n, Cin, Cout, Tout, Kw, stride, dil, ptr(ws), ws.numel(),
                                          st), 'w2l_conv1d_igemm_ws')
        return
    assert y.dtype == torch.float32
    check(lib.w2l_conv1d_igemm(xptr(x.hi), bstride, rows_total, ptr(w_hi), ptr(y), 1, 0, ptr(bias), None, n, Cin, Cout,
                               Tout, Kw, stride, dil, st), 'w2l_conv1d_igemm')
    check(lib.w2l_conv1d_igemm(xptr(x.hi), bstride, rows_total, ptr(w_lo), ptr(y), 1, 1, None, None, n, Cin, Cout, Tout,
                               Kw, stride, dil, st), 'w2l_conv1d_igemm')
    check(lib.w2l_conv1d_igemm(xptr(x.lo), bstride, rows_total, ptr(w_hi), ptr(y), 1, 1, None, ptr(stats), n, Cin, Cout,
                               Tout, Kw, stride, dil, st), 'w2l_conv1d_igemm')


# fp8 mode (BASELINE config 5): per-tensor scales of the e4m3 operands.  Activations: clamp(0, 20) outputs times 16 stay
# below e4m3's 448; ReLU / linear outputs (BatchNorm-normalised, a few units wide) times 8, saturating beyond 56.  Weights:
# the largest power of two that keeps |w| * scale <= 448, re-derived from the tensor's amax every FP8_WEIGHT_RESCALE
# weight versions without a host sync (_fp8_weights: device-side amax, asynchronous copy, adopted FP8_RESCALE_LAG versions later).
FP8_ACT_SCALE = {ACT_CLAMP20: 16.0, ACT_RELU: 8.0, ACT_NONE: 8.0}
FP8_WEIGHT_RESCALE = 256
FP8_RESCALE_LAG = 8                # weight versions between asking for the amax and adopting the scale it yields: the copy has
                                   # landed many steps earlier even when the host runs ahead of the GPU, so the event wait
                                   # below never drains the pipeline, and every data-parallel rank adopts at the same step
AMAX_SLOTS = 64                    # include/w2l_hip.h W2L_AMAX_SLOTS
# fp8 mode, data gradients: on e4m3 operands too ('1'), in bf16 ('0'), or (default 'auto') e4m3 only from FP8_DGRAD_MIN_ROWS
# rows of dy per launch.  The e4m3 data gradient needs two more launches per layer on the backward critical path (dy's
# quantisation, and the BatchNorm-backward reduction its epilogue cannot form) -- measured on one MI355X, ms per step,
# bf16 / fp8 forward only / fp8 forward + data gradient: Wav2Letter N=32 x T=1000 13.76 / 12.7-12.9 / 12.3-12.4, Jasper 10x5
# N=16 x T=1000 19.4 / 18.4 / 18.75, Jasper 10x5 N=16 x T=16000 208 / 190 / 170: a gain from ~12 000 rows per launch.
# With the weight gradients on e4m3 operands as well (one quantisation pass of dy serves both, and the main and the side
# stream get shorter together -- either alone leaves the step on the other stream's timeline: Jasper 10x5 N=16 x T=1000 17.96
# with neither, 17.83 data gradients only, 17.48 weight gradients only, 15.18 with both) the gain starts far lower: Wav2Letter
# N=16 8.01 -> 6.61, N=8 (4 000 rows) 5.56 -> 5.34.
FP8_DGRAD = os.environ.get('W2L_FP8_DGRAD', 'auto')
FP8_DGRAD_MIN_ROWS = 3072
# fp8 mode, weight gradients: '1' = on e4m3 operands too (w2l_conv1d_wgrad_fp8: dy's e4m3 copy x the e4m3 copy of the input
# the forward convolution already consumed), '0' = bf16, 'auto' (default) = e4m3 from FP8_DGRAD_MIN_ROWS rows, like the data
# gradients (dy's quantisation pass is then shared by the two)
FP8_WGRAD = os.environ.get('W2L_FP8_WGRAD', 'auto')
JOIN_EVENTS = None                 # a list: backward() appends (event on the main stream, event on the weight-gradient stream) at its join


_fp8_epoch = [0]                   # bumped whenever a weight adopts a new e4m3 scale (replay.py: records made before are stale)


def _pow2_scale(amax: float) -> float:
    """the largest power of two s with amax * s <= 448 (e4m3's largest finite value)"""
    return 2.0 ** math.floor(math.log2(448.0 / max(float(amax), 1e-30)))


def _fp8_weights(conv: ConvSpec, pk: '_PackedW', dgrad: bool = False):
    """(e4m3 operand, scale) of a conv weight, requantised from the bf16 pack when the weight changed: the forward layout
    [Kw, CoutP, CinP], or (``dgrad``) the flipped-tap layout [Kw, CinP, CoutP] of the data gradient -- one scale for both.

    The scale is a host float (the kernels take it by value).  It is derived from the tensor's amax with a host sync ONCE,
    at first use; afterwards every FP8_WEIGHT_RESCALE weight versions the amax is re-taken on the device and copied to pinned
    host memory asynchronously, and the scale it yields is adopted FP8_RESCALE_LAG versions later (by then the copy has long
    landed: no pipeline drain, and every rank of a data-parallel run adopts it at the same step).  Weights move by lr * grad
    per step and the conversion saturates, so a scale that lags by a few steps is harmless.  Under stream capture
    (graph.GraphedTrainStep) the scale in force when the graph was captured stays."""
    w = conv.weight
    st = w.__dict__.get('_w2l_fp8')
    if st is None or st['q'].shape != pk.fwd_hi.shape or st['q'].device != pk.fwd_hi.device:
        _lib.poison('fp8 weight scale: first use')
        mn, mx = torch.aminmax(pk.fwd_hi)
        scale = _pow2_scale(max(-float(mn), float(mx)))          # host sync: first use only
        dev = pk.fwd_hi.device
        st = _Volatile({'q': torch.empty(pk.fwd_hi.shape, dtype=torch.uint8, device=dev),
                        'qd': torch.empty(pk.dgr_hi.shape, dtype=torch.uint8, device=dev), 'scale': scale, 'age': 0,
                        'version': None, 'version_d': None, 'req': None})
        w.__dict__['_w2l_fp8'] = st
    elif not dgrad and not torch.cuda.is_current_stream_capturing():
        req = st.get('req')
        if req is not None and st['age'] >= FP8_RESCALE_LAG:
            ev, host = req
            ev.synchronize()                                         # FP8_RESCALE_LAG versions old: returns at once
            _lib.poison('fp8 weight scale: adoption point')
            st['req'] = None
            scale = _pow2_scale(max(-float(host[0]), float(host[1])))
            if scale != st['scale']:
                st['scale'] = scale
                _fp8_epoch[0] += 1                                   # recorded launch lists carry the old scale by value
                st['version'] = st['version_d'] = None               # both layouts are requantised below / by the data gradient
        elif req is None and st['age'] >= FP8_WEIGHT_RESCALE:
            _lib.poison('fp8 weight scale: amax request')
            host = torch.empty(2, dtype=torch.float32, pin_memory=True)
            host.copy_(torch.stack(torch.aminmax(pk.fwd_hi)).float(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st['req'] = (ev, host)
            st['age'] = 0
    key, vkey, src = ('qd', 'version_d', pk.dgr_hi) if dgrad else ('q', 'version', pk.fwd_hi)
    if st[vkey] != pk.version:
        check(lib.w2l_quantize_e4m3(ptr(src), 0, src.numel(), st['scale'], ptr(st[key]), stream_ptr()), 'w2l_quantize_e4m3')
        st[vkey] = pk.version
        if not dgrad:
            st['age'] += 1
    return st[key], st['scale']


class StackEngine:
    """Executes a list of UnitSpec.  ``precise`` selects the split-bf16 (near-fp32) mode used
    for parity against the fp32 reference; the default is bf16 operands with fp32 accumulate; ``fp8`` runs the forward
    and data-gradient convolutions of the units on e4m3 operands (activations, dy and weights quantised per tensor, fp32
    accumulate) while the classifier, the weight gradients and all statistics stay as in bf16 mode."""

    def __init__(self, units: Sequence[UnitSpec], head: Optional[ConvSpec], n_labels: int, precise: bool = False,
                 fp8: bool = False):
        if fp8 and precise:
            raise ValueError('fp8 operands and the fp32 parity mode exclude each other')
        self.fp8 = fp8
        """``head`` = the classifier conv followed by (log_)softmax; ``None`` runs an OPEN stack whose result is the last
        unit's activation as fp32 [N, C, T'] (stand-alone Conv1dBlock / MaskedConv1d / JasperBlock modules)."""
        self.units = list(units)
        self.head = head
        self.n_labels = n_labels
        self.precise = precise
        # callbacks for data-parallel overlap (distributed.GradReducer): grad_ready(param, grad, dense_storage)
        self.grad_ready: Optional[Callable] = None
        self.flat_ready: Optional[Callable] = None        # flat_ready(buffer): every per-channel gradient of the step, one message
        self.backward_done: Optional[Callable[[], None]] = None
        # weight gradients are off the backward critical path (only dgrad feeds the next layer): they run
        # on a side HIP stream so their blocks fill the tail rounds of the dgrad / elementwise kernels
        self.overlap_wgrad = True
        self._side = None
        self._main_stream = None     # the caller's stream, looked up once per backward (torch.cuda.current_stream is not free)
        self._side_used = False
        self._held: list = []        # tensors in use by side-stream kernels; released after the join in backward()
        self._dyq: dict = {}         # fp8 mode: id(dy) -> (e4m3 copy, 1/scale) shared by a unit's weight and data gradient
        self._nbt_pending: list = []  # num_batches_tracked buffers to bump (one fused launch per forward)
        self._zero_pool = None       # (buffer, offset): the identically-zero conv-bias gradients of one backward
        # graph mode (graph.GraphedTrainStep): a uint64 step counter in device memory; dropout offsets become
        # (unit index) + counter, so a captured step draws new masks at every replay
        self.dropout_counter: Optional[torch.Tensor] = None
        # fp8 mode: device counter of activation elements whose e4m3 copy saturated (fixed per-tensor activation scales:
        # clamp(0, 20) outputs never do, ReLU outputs beyond 448 / 8 = 56 would) -- read on demand by fp8_saturated()
        self._q_clipped: Optional[torch.Tensor] = None
        # deferred weight gradients (optim.FusedSGD.defer_wgrad): the weight gradients of the TOP ``defer_wgrad`` units --
        # computed first in backward, needed last by the next forward -- are not launched in backward but at the START of
        # the next forward, on the weight-gradient stream, each followed by its fused update (``deferred.apply``) and, data
        # parallel, preceded by its all-reduce (``grad_reduce_start``).  They then run beside the forward's BatchNorm chain
        # and CTC, where no other MFMA-bound kernel is ready (profiles/r03_step_timeline.txt: 1.5 ms per step).
        # An int k = the top k units; a set of unit indices = exactly those (e.g. every other one of the top layers, so that
        # the backward pass keeps weight gradients of its own to run beside its BatchNorm kernels).
        self.defer_wgrad = 0
        self.deferred = None              # the optimizer: accepts(p), token(), stepped(token), apply(p, grad)
        self.grad_reduce_start: Optional[Callable] = None      # distributed.GradReducer.start
        self._deferred: list = []
        # grouped weight gradients (wgrad_groups.py): id(conv) -> (group id, members) for this backward pass, and the members
        # of each group that have arrived so far -- a group is launched when its last member's dy exists
        self._wg_of: dict = {}
        self._wg_pending: dict = {}

    # ------------------------------------------------------------------ parameters
    def parameters(self) -> List[torch.Tensor]:
        out = []
        for u in self.units:
            if u.dw is not None:
                out += u.dw.params()
            out += u.main.params()
            if u.res is not None:
                out += u.res.params()
        if self.head is not None:
            out += self.head.params()
        return out

    # ------------------------------------------------------------------ forward
    def _in_pad_for(self, act_index: int):
        """padding the consumers of activation ``act_index`` need (max over consumers)"""
        pl = pr = 0
        mode = PAD_ZERO
        convs = []
        for u in self.units:
            if u.src == act_index:
                convs.append(u.dw if u.dw is not None else u.main)
            if u.res is not None and u.res_src == act_index:
                convs.append(u.res)
        if act_index == len(self.units) and self.head is not None:
            convs.append(self.head)
        for c in convs:
            if c.pad_l > pl or c.pad_r > pr:
                pl, pr = max(pl, c.pad_l), max(pr, c.pad_r)
            if c.pad_l or c.pad_r:
                mode = c.pad_mode
        return pl, pr, mode

    def forward(self, x: torch.Tensor, lens: Optional[torch.Tensor], training: bool, softmax_mode: int = 0,
                want_input_grad: bool = False):
        """x fp32 [N, C, T] on device -> (out fp32 [N, T', n_labels], lens_out or None)."""
        _in_forward[0] = True
        try:
            return self._forward(x, lens, training, softmax_mode, want_input_grad)
        finally:
            _in_forward[0] = False
            lib.w2l_conv_stats_mode(0)      # thread-local library state: never left in slot mode, whatever was raised
            self._stat_pool = None

    def _forward(self, x, lens, training, softmax_mode, want_input_grad):
        _lib.require_device(x)
        global _dropout_calls
        if self._deferred:
            self.flush_deferred(pos=0)
        precise = self.precise
        N, C0, T0 = x.shape
        # folded statistics finalize: the convolutions of this forward add their statistics onto STAT_SLOTS rows of ONE
        # zero-filled pool (one fill launch per forward, not one per layer)
        fold = training and not precise and x.is_cuda and not DETERMINISTIC_WGRAD and (
            FOLD_BN_FWD in ('1', True) or (FOLD_BN_FWD == 'auto' and self.units and
                                           N * T0 // max(1, self.units[0].main.stride) < FOLD_BN_FWD_MAX_ROWS))
        self._stat_pool = None
        self._after_apply = []
        if fold:
            need = 0
            for u in self.units:
                for c in (u.main, u.res):
                    if c is not None and c.has_bn:
                        need += STAT_SLOTS * 2 * padded_channels(c.cout)
            fold = need > 0
            if fold:
                self._stat_pool = [zeros(need, torch.float32, x.device), 0]
        lib.w2l_conv_stats_mode(STAT_SLOTS if fold else 0)        # (thread-local: this thread's launches)
        x = x.contiguous().float()
        dev = x.device
        st = stream_ptr
        ctx = {'units': [], 'acts': [], 'training': training, 'x_shape': (N, C0, T0), 'want_dx': bool(want_input_grad)}
        lens_dev, mid_lens, out_lens, lens_final = self._plan_lens(lens, dev)
        # ---- activation 0: the spectrogram, channels-last, padded for its consumers
        pl, pr, mode = self._in_pad_for(0)
        cp0 = padded_channels(C0)
        a_hi = torch.empty(N, pl + T0 + pr, cp0, dtype=torch.bfloat16, device=dev)
        a_lo = torch.empty_like(a_hi) if precise else None
        # the input is masked by its lengths only if a MaskedConv1d reads it (a block of plain convolutions sees it whole)
        if lens_dev is not None and not any((u.src == 0 and u.update_lens) for u in self.units):
            in_mask = None
        else:
            in_mask = lens_dev
        check(lib.w2l_nct_to_ntc(ptr(x), N, C0, T0, cp0, pl, pr, mode, ptr(in_mask), ptr(a_hi), ptr(a_lo), st()),
              'w2l_nct_to_ntc')
        acts: List[Act] = [Act(a_hi, a_lo, N, T0, C0, cp0, pl, pr, mode, in_mask)]
        if self.fp8 and self.head is None and self.units and cp0 % 128 == 0 and self._fp8_consumers(0):
            # an OPEN stack in fp8 mode (a Conv1dBlock / JasperBlock called on its own): inside a network the block's input
            # is the previous block's activation and carries an e4m3 copy at that activation's scale; give the caller's
            # tensor the same, so that the block's first convolutions run on e4m3 operands as they do in the network
            a0 = acts[0]
            a0.q = torch.empty(a_hi.shape, dtype=torch.uint8, device=dev)
            a0.q_scale = FP8_ACT_SCALE[self.units[0].act]
            check(lib.w2l_quantize_e4m3(ptr(a_hi), 0, a_hi.numel(), a0.q_scale, ptr(a0.q), st()), 'w2l_quantize_e4m3')

        for ui, u in enumerate(self.units):
            if ui and self._deferred:       # held-back weight gradients due in front of this unit (flush_deferred)
                self.flush_deferred(pos=ui)
            uc = _UnitCtx(unit=u)
            src = acts[u.src]
            conv = u.main
            if u.dw is not None:            # depthwise conv -> (masked) intermediate activation -> 1x1 pointwise conv
                dwc = u.dw
                src = self._dw_forward(dwc, src, mid_lens[ui])
                uc.mid = src
            y, stats, Tout = self._conv_forward(conv, src, need_stats=conv.has_bn and training)
            uc.y, uc.Tout = y, Tout
            coutp = y.shape[2]
            fins = [None, None]             # folded finalize records (w2l_bn_act_fwd_fin) of the two branches
            if conv.has_bn:
                if fold:
                    fins[0], (uc.scale, uc.shift, uc.mean, uc.invstd) = self._bn_fin_record(conv, stats, N * Tout, coutp)
                else:
                    uc.scale, uc.shift, uc.mean, uc.invstd = self._bn_finalize(conv, stats, N * Tout, coutp, training)
            if u.res is not None:
                rsrc = acts[u.res_src]
                y2, stats2, Tout2 = self._conv_forward(u.res, rsrc, need_stats=u.res.has_bn and training)
                if Tout2 != Tout or y2.shape != y.shape:
                    raise ValueError('residual branch shape mismatch')
                uc.y2 = y2
                if u.res.has_bn:
                    if fold:
                        fins[1], (uc.scale2, uc.shift2, uc.mean2, uc.invstd2) = self._bn_fin_record(u.res, stats2, N * Tout, coutp)
                    else:
                        uc.scale2, uc.shift2, uc.mean2, uc.invstd2 = self._bn_finalize(u.res, stats2, N * Tout, coutp, training)
            uc.lens_out = out_lens[ui]               # length mask of this unit's output (None: not masked)
            # ---- BN-apply + dropout + activation -> padded input of the next conv
            opl, opr, omode = self._in_pad_for(ui + 1)
            out_hi = torch.empty(N, opl + Tout + opr, coutp, dtype=torch.bfloat16, device=dev)
            out_lo = torch.empty_like(out_hi) if precise else None
            p = u.drop_p if training else 0.0
            if p > 0.0:
                uc.mask = torch.empty(N * Tout * (coutp // 8), dtype=torch.uint8, device=dev)
                uc.seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
                if self.dropout_counter is not None:
                    uc.offset = ui + 1
                else:
                    _dropout_calls += 1
                    uc.offset = _dropout_calls
            d = self._desc(uc, N, Tout, coutp, p, uc.lens_out)
            out_q, q_scale = None, 1.0
            if self.fp8 and coutp % 128 == 0 and self._fp8_consumers(ui + 1):
                out_q = torch.empty(N, opl + Tout + opr, coutp, dtype=torch.uint8, device=dev)
                q_scale = FP8_ACT_SCALE[u.act]
                if self._q_clipped is None or self._q_clipped.device != dev:
                    _lib.poison('fp8 saturation counter created')
                    self._q_clipped = torch.zeros(1, dtype=torch.int64, device=dev)
                d.q_clipped = self._q_clipped.data_ptr()
            if fins[0] is not None or fins[1] is not None:
                f1 = fins[0] if fins[0] is not None else _lib.BnFin()          # (partial NULL: scale / shift from the descriptor)
                f2 = (fins[1] if fins[1] is not None else _lib.BnFin()) if u.res is not None else None
                check(lib.w2l_bn_act_fwd_fin(C.byref(d), C.byref(f1), C.byref(f2) if f2 is not None else None, ptr(out_hi),
                                             ptr(out_q), q_scale, opl + Tout + opr, opl, opr, omode, st()), 'w2l_bn_act_fwd_fin')
                for rm, rv, rm_p, rv_p in self._after_apply:       # running statistics of a channel count that is padded
                    _lib.poison('running statistics of a padded channel count')
                    rm.copy_(rm_p[: rm.numel()])
                    rv.copy_(rv_p[: rv.numel()])
                self._after_apply = []
            else:
                check(lib.w2l_bn_act_fwd_q(C.byref(d), ptr(out_hi), ptr(out_lo), ptr(out_q), q_scale, opl + Tout + opr, opl, opr,
                                           omode, st()), 'w2l_bn_act_fwd_q')
            uc.out_index = ui + 1
            acts.append(Act(out_hi, out_lo, N, Tout, conv.cout, coutp, opl, opr, omode, uc.lens_out, out_q, q_scale))
            ctx['units'].append(uc)

        if self._nbt_pending:
            self._bump_counters(self._nbt_pending)
            self._nbt_pending = []
        if fold:
            lib.w2l_conv_stats_mode(0)
            self._stat_pool = None
        ctx['acts'] = acts
        ctx['softmax_mode'] = softmax_mode
        ctx['lens_out'] = lens_final
        last = acts[-1]
        if self.head is None:           # open stack: hand the last activation back in the caller's layout (cold path, torch ops)
            _lib.poison('open stack')
            a = last.hi[:, last.pad_l:last.pad_l + last.T, :last.C].float()
            if last.lo is not None:
                a = a + last.lo[:, last.pad_l:last.pad_l + last.T, :last.C].float()
            ctx['out'] = a.transpose(1, 2).contiguous()
            return ctx['out'], ctx
        # ---- classifier (1x1 conv, bias, no BN) + (log_)softmax
        logits, _, Th = self._conv_forward(self.head, last, need_stats=False, force_f32=True)
        out = torch.empty(N, Th, self.n_labels, dtype=torch.float32, device=dev)
        check(lib.w2l_log_softmax_fwd(ptr(logits), N, Th, self.n_labels, logits.shape[2], softmax_mode, ptr(out), st()),
              'w2l_log_softmax_fwd')
        ctx['out'] = out
        for hook in list(AFTER_FORWARD):
            _py(hook)
        return out, ctx

    def _bump_counters(self, tensors):
        """num_batches_tracked += 1 of every BatchNorm1d that ran in training mode (wav2letter.py:37), one launch: a device
        table of the counters' addresses, built once per set of counters"""
        if not tensors[0].is_cuda or any(t.dtype != torch.int64 for t in tensors):
            _lib.poison('BatchNorm counters off the device')
            torch._foreach_add_(tensors, 1)
            return
        key = tuple(t.data_ptr() for t in tensors)
        hit = self.__dict__.get('_nbt_table')
        if hit is None or hit[0] != key:
            _lib.poison('BatchNorm counter table built')
            table = torch.tensor(key, dtype=torch.int64).to(tensors[0].device)
            hit = self.__dict__['_nbt_table'] = (key, table, list(tensors))
        check(lib.w2l_add_i64_multi(ptr(hit[1]), len(key), 1, stream_ptr()), 'w2l_add_i64_multi')

    def fp8_saturated(self, reset: bool = True) -> int:
        """fp8 mode: how many activation elements saturated the e4m3 range (|a| * scale > 448) since the last reset.
        SYNCHRONISES (reads a device counter): call it at a logging point, not per step."""
        if self._q_clipped is None:
            return 0
        n = int(self._q_clipped.item())
        if reset and n:
            self._q_clipped.zero_()
        return n

    def _fp8_consumers(self, act_index: int) -> bool:
        """does a stride-1 dense conv of some unit read this activation (the e4m3 copy is written only then)?"""
        for u in self.units:
            if u.src == act_index and u.dw is None and u.main.stride == 1:
                return True
            if u.res is not None and u.res_src == act_index and u.res.stride == 1:
                return True
        return False

    def _plan_lens(self, lens, dev):
        """Length bookkeeping of the MaskedConv1d chain (jasper.py:109-121): lens <- (lens + 2p - d(k-1) - 1) / s + 1 in
        float (true division), truncated to integers where a mask is applied.  Returns (int32 device lengths of the input,
        per-unit lengths after the depthwise conv, per-unit output mask lengths, final float lengths).

        With host lengths (what _collator hands over, data_loader.py:150) the whole chain is evaluated on the host with the
        reference's own torch ops and uploaded in ONE copy; device lengths take the same arithmetic as device ops (a few tiny
        launches per conv, and the caller pays a sync to read the result back)."""
        n_units = len(self.units)
        mid, outl = [None] * n_units, [None] * n_units
        if lens is None or not any(u.update_lens or u.mask_out for u in self.units):
            return None, mid, outl, None
        on_host = not lens.is_cuda
        if on_host:
            # host lengths: the chain in numpy float32 (same operations, same order as the device branch below), ONE upload --
            # into the record set's static table while a step is being recorded / replayed (replay.py)
            from .replay import lens_rows_host
            rows, mid_has, out_has, final = lens_rows_host(self, lens)
            st = getattr(self, '_lens_static', None)
            if st is not None:
                st['host'].numpy()[...] = rows
                st['dev'].copy_(st['host'], non_blocking=True)
                st['event'].record()
                packed = st['dev']
            else:
                packed = torch.from_numpy(rows).pin_memory().to(dev, non_blocking=True)
            it = iter(packed.unbind(0))
            first_d = next(it)
            mid = [next(it) if h else None for h in mid_has]
            outl = [next(it) if h else None for h in out_has]
            return first_d, mid, outl, final
        _lib.poison('lengths on the device')
        cur = lens.to(device=dev, dtype=torch.int32)
        first = cur
        cur_f = cur.float()
        for ui, u in enumerate(self.units):
            if u.dw is not None and u.update_lens:
                c = u.dw
                cur_f = (cur_f + (c.pad_l + c.pad_r) - c.dilation * (c.kernel - 1) - 1) / c.stride + 1
                cur = cur_f.to(torch.int32)
                mid[ui] = cur
            if u.update_lens:
                c = u.main
                cur_f = (cur_f + (c.pad_l + c.pad_r) - c.dilation * (c.kernel - 1) - 1) / c.stride + 1
                cur = cur_f.to(torch.int32)
            if u.mask_out:
                outl[ui] = cur
        return first, mid, outl, cur_f

    def _conv_forward(self, conv: ConvSpec, src: Act, need_stats: bool, force_f32: bool = False):
        if src.pad_l < conv.pad_l or src.pad_r < conv.pad_r:
            raise ValueError('activation buffer is not padded enough for its consumer')
        pk = pack_weights(conv, self.precise)
        if pk.cinp != src.CP:
            raise ValueError(f'channel mismatch: conv expects {pk.cinp} padded channels, activation has {src.CP}')
        N = src.N
        Tp = src.T + conv.pad_l + conv.pad_r
        Tout = (Tp - (conv.kernel - 1) * conv.dilation - 1) // conv.stride + 1
        if Tout <= 0:
            raise ValueError('input too short for this convolution')
        f32 = self.precise or force_f32
        y = torch.empty(N, Tout, pk.coutp, dtype=torch.float32 if f32 else torch.bfloat16, device=src.hi.device)
        stats = None
        if need_stats:
            pool = getattr(self, '_stat_pool', None)
            if pool is not None:            # folded finalize: STAT_SLOTS zero rows of the forward's pool, added to atomically
                n = STAT_SLOTS * 2 * pk.coutp
                stats = pool[0][pool[1]: pool[1] + n].view(STAT_SLOTS, 2, pk.coutp)
                pool[1] += n
            else:
                tiles = lib.w2l_conv_stat_tiles(N, Tout)
                stats = torch.empty(tiles, 2, pk.coutp, dtype=torch.float32, device=src.hi.device)
        bias = _padded_vec(conv.bias, pk.coutp, 0.0)
        flops = 2.0 * N * Tout * conv.cout * conv.cin * conv.kernel
        if self.fp8 and src.q is not None and conv.stride == 1 and pk.cinp % 128 == 0 and not force_f32:
            self._igemm_fp8(conv, pk, src, y, bias, stats, Tout, flops)
        else:
            _igemm(src, src.pad_l - conv.pad_l, pk.fwd_hi, pk.fwd_lo, y, bias, stats, pk.cinp, pk.coutp, Tout, conv.kernel,
                   conv.stride, conv.dilation, self.precise, alg_flops=flops)
        return y, stats, Tout

    def _igemm_fp8(self, conv: ConvSpec, pk: _PackedW, src: Act, y, bias, stats, Tout, flops):
        """forward conv on e4m3 operands (w2l_conv1d_igemm_fp8): y = (xq * wq) / (x scale * w scale) + bias"""
        wq, w_scale = _fp8_weights(conv, pk)
        N, cin, cout = src.N, pk.cinp, pk.coutp
        row_off = src.pad_l - conv.pad_l
        xq = C.c_void_p(src.q.data_ptr() + row_off * src.CP)
        bstride, rows_total = src.rows * src.CP, N * src.rows - row_off
        y_f32 = int(y.dtype == torch.float32)
        st = stream_ptr()
        if AUTOTUNE:
            key = ('fp8', N, cin, cout, Tout, conv.kernel, conv.dilation, stats is not None, src.q.device.index)
            if key not in _tuned_shapes:
                _tuned_shapes.add(key)
                _tune_state['dirty'] = True
                check(lib.w2l_conv1d_igemm_fp8_tune(xq, bstride, rows_total, ptr(wq), ptr(y), y_f32, ptr(bias), ptr(stats), N,
                                                    cin, cout, Tout, conv.kernel, conv.dilation, TUNE_REPS, st),
                      'w2l_conv1d_igemm_fp8_tune')
                if stats is not None and getattr(self, '_stat_pool', None) is not None:
                    stats.zero_()                                   # (the measuring launches ADDED to the statistics rows)
        with _timed('conv_igemm_fp8_kernel', flops):
            check(lib.w2l_conv1d_igemm_fp8(xq, bstride, rows_total, ptr(wq), ptr(y), y_f32, 1.0 / (src.q_scale * w_scale), None,
                                           ptr(bias), ptr(stats), N, cin, cout, Tout, conv.kernel, conv.dilation, st),
                  'w2l_conv1d_igemm_fp8')

    @staticmethod
    def _dw_weight(dwc: ConvSpec, cp: int) -> torch.Tensor:
        """fp32 tap-major [K][CP] view / padded copy of a depthwise weight (logical [C, 1, K])"""
        w = dwc.weight.detach()
        c, _, k = w.shape
        km = w.permute(2, 0, 1).reshape(k, c)            # no copy for the tap-major parameter layout
        if cp == c and km.is_contiguous():
            return km
        _lib.poison('depthwise weight of a padded channel count')
        out = torch.zeros(k, cp, dtype=torch.float32, device=w.device)
        out[:, :c] = km
        return out

    def _dw_forward(self, dwc: ConvSpec, src: Act, lens_mid) -> Act:
        if src.pad_l < dwc.pad_l or src.pad_r < dwc.pad_r:
            raise ValueError('activation buffer is not padded enough for its depthwise consumer')
        N, cp = src.N, src.CP
        Tp = src.T + dwc.pad_l + dwc.pad_r
        Tmid = (Tp - (dwc.kernel - 1) * dwc.dilation - 1) // dwc.stride + 1
        dev = src.hi.device
        mid_hi = torch.empty(N, Tmid, cp, dtype=torch.bfloat16, device=dev)
        mid_lo = torch.empty_like(mid_hi) if self.precise else None
        row_off = src.pad_l - dwc.pad_l
        off = row_off * cp * 2
        w = self._dw_weight(dwc, cp)
        check(lib.w2l_dwconv_fwd(C.c_void_p(src.hi.data_ptr() + off),
                                 C.c_void_p(src.lo.data_ptr() + off) if src.lo is not None else None, src.rows, ptr(w),
                                 ptr(mid_hi), ptr(mid_lo), N, Tmid, cp, dwc.kernel, dwc.stride, dwc.dilation, ptr(lens_mid),
                                 stream_ptr()), 'w2l_dwconv_fwd')
        return Act(mid_hi, mid_lo, N, Tmid, dwc.cout, cp, 0, 0, PAD_ZERO, lens_mid)

    def _dw_backward(self, dwc: ConvSpec, dmid, src: Act, mid: Act, need_dx: bool, grads):
        """depthwise weight gradient (+ data gradient) from the gradient wrt the pointwise conv's input"""
        g, _, _, _, per = dmid[:5]
        N, cp = src.N, src.CP
        dev = g.device
        row_off = src.pad_l - dwc.pad_l
        off = row_off * cp * 2
        k = dwc.kernel
        dwg = zeros((k, cp), torch.float32, dev)
        check(lib.w2l_dwconv_wgrad(ptr(g), int(g.dtype == torch.float32), per, C.c_void_p(src.hi.data_ptr() + off),
                                   C.c_void_p(src.lo.data_ptr() + off) if src.lo is not None else None, src.rows, ptr(dwg), N,
                                   mid.T, cp, k, dwc.stride, dwc.dilation, ptr(mid.lens), stream_ptr()), 'w2l_dwconv_wgrad')
        c = dwc.cout
        gw = dwg.view(k, cp, 1).permute(1, 2, 0)          # logical [CP, 1, K] in the parameter's tap-major layout
        self._set(grads, dwc.weight, gw if cp == c else gw[:c], storage=dwg)
        if not need_dx:
            return None
        Tp = src.T + dwc.pad_l + dwc.pad_r
        tmid, lens_mid = mid.T, mid.lens
        if dwc.stride != 1:                # cold path (spectrogram gradient only): zero-stuff to stride 1
            _lib.poison('strided depthwise data gradient')
            s_ = dwc.stride
            tup = (mid.T - 1) * s_ + 1
            up = torch.zeros(N, tup, cp, dtype=g.dtype, device=dev)
            up[:, 0:tup:s_] = g.view(N, per, cp)[:, :mid.T]
            g, per, tmid = up, tup, tup
            if lens_mid is not None:
                lens_mid = ((lens_mid - 1) * s_ + 1).clamp(min=0).to(torch.int32)
        dxp = torch.empty(N, Tp, cp, dtype=torch.float32 if self.precise else torch.bfloat16, device=dev)
        w = self._dw_weight(dwc, cp)
        check(lib.w2l_dwconv_dgrad(ptr(g), int(g.dtype == torch.float32), per, ptr(w), ptr(dxp),
                                   int(dxp.dtype == torch.float32), N, Tp, tmid, cp, k, dwc.dilation, ptr(lens_mid),
                                   stream_ptr()), 'w2l_dwconv_dgrad')
        return (dxp, dwc.pad_l, dwc.pad_r, dwc.pad_mode, Tp)

    def _bn_finalize(self, conv: ConvSpec, stats, count, cp, training):
        dev = conv.weight.device
        scale, shift, mean, invstd = torch.empty(4, cp, dtype=torch.float32, device=dev).unbind(0)     # one allocation
        gamma = _padded_vec(conv.bn_weight, cp, 1.0)
        beta = _padded_vec(conv.bn_bias, cp, 0.0)
        rm, rv = conv.running_mean, conv.running_var
        padded_running = rm is not None and rm.numel() != cp
        if padded_running:
            rm_p, rv_p = _padded_vec(rm, cp, 0.0).clone(), _padded_vec(rv, cp, 1.0).clone()
        else:
            rm_p, rv_p = rm, rv
        if training:
            ntiles = stats.shape[0]
            check(lib.w2l_bn_finalize(ptr(stats), ntiles, cp, count, ptr(gamma), ptr(beta), conv.eps, conv.momentum,
                                      ptr(rm_p), ptr(rv_p), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                      stream_ptr()), 'w2l_bn_finalize')
            if padded_running:
                _lib.poison('running statistics of a padded channel count')
                rm.copy_(rm_p[: rm.numel()])
                rv.copy_(rv_p[: rv.numel()])
            if conv.num_batches_tracked is not None:
                self._nbt_pending.append(conv.num_batches_tracked)     # one foreach add at the end of forward
        else:
            check(lib.w2l_bn_finalize(None, 0, cp, 1, ptr(gamma), ptr(beta), conv.eps, conv.momentum, ptr(rm_p),
                                      ptr(rv_p), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), stream_ptr()),
                  'w2l_bn_finalize')
        return scale, shift, mean, invstd

    def _bn_fin_record(self, conv: ConvSpec, stats, count, cp):
        """the finalize of one BatchNorm branch as a record for w2l_bn_act_fwd_fin (training mode, STAT_SLOTS statistics rows);
        returns (record, (scale, shift, mean, invstd)) -- the four vectors are written by that launch"""
        dev = conv.weight.device
        scale, shift, mean, invstd = torch.empty(4, cp, dtype=torch.float32, device=dev).unbind(0)     # one allocation
        gamma = _padded_vec(conv.bn_weight, cp, 1.0)
        beta = _padded_vec(conv.bn_bias, cp, 0.0)
        rm, rv = conv.running_mean, conv.running_var
        if rm is not None and rm.numel() != cp:
            rm_p, rv_p = _padded_vec(rm, cp, 0.0).clone(), _padded_vec(rv, cp, 1.0).clone()
            self._after_apply.append((rm, rv, rm_p, rv_p))
        else:
            rm_p, rv_p = rm, rv
        f = _lib.BnFin()
        f.partial, f.rows, f.count = stats.data_ptr(), stats.shape[0], count
        f.gamma, f.beta = gamma.data_ptr(), beta.data_ptr()
        f.eps, f.momentum = conv.eps, conv.momentum
        f.running_mean = rm_p.data_ptr() if rm_p is not None else None
        f.running_var = rv_p.data_ptr() if rv_p is not None else None
        f.mean, f.invstd, f.scale, f.shift = mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr()
        f._keep = (gamma, beta, rm_p, rv_p, stats)          # alive until the launch has been enqueued (same stream: enough)
        if conv.num_batches_tracked is not None:
            self._nbt_pending.append(conv.num_batches_tracked)
        return f, (scale, shift, mean, invstd)

    def _desc(self, uc: _UnitCtx, N, T, cp, p, lens, constant_stats: bool = False) -> BnActDesc:
        """``constant_stats``: leave mean / invstd out, which makes w2l_bn_act_bwd_apply treat BatchNorm as the per-channel
        affine map it is in eval mode (dy = scale * g)"""
        d = BnActDesc()
        d.N, d.T, d.C = N, T, cp
        d.y = uc.y.data_ptr()
        d.y_f32 = int(uc.y.dtype == torch.float32)
        for name in ('scale', 'shift', 'mean', 'invstd', 'scale2', 'shift2', 'mean2', 'invstd2'):
            t = getattr(uc, name)
            if constant_stats and name in ('mean', 'invstd', 'mean2', 'invstd2'):
                t = None
            setattr(d, name, t.data_ptr() if t is not None else None)
        d.y2 = uc.y2.data_ptr() if uc.y2 is not None else None
        d.act = uc.unit.act
        d.drop_p = float(p)
        d.seed, d.offset = uc.seed, uc.offset
        d.mask = uc.mask.data_ptr() if uc.mask is not None else None
        d.lens = lens.data_ptr() if lens is not None else None
        d.offset_dev = self.dropout_counter.data_ptr() if (self.dropout_counter is not None and p > 0.0) else None
        uc.keep.append(lens)
        return d

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, g_out: torch.Tensor):
        """g_out: gradient wrt the (log_)softmax output [N, T', n_labels].  Returns the list of
        parameter gradients in ``self.parameters()`` order."""
        try:
            return self._backward(ctx, g_out)
        finally:
            lib.w2l_conv_stats_mode(0)      # (thread-local: this -- the autograd -- thread's)
            self._slot_pool = None

    def _backward(self, ctx, g_out: torch.Tensor):
        acts: List[Act] = ctx['acts']
        out = ctx['out']
        N = out.shape[0]
        dev = out.device
        grads = {}
        # eval-mode forward normalised with the RUNNING statistics: they are constants of the step, so dy = scale * g (no
        # batch-mean terms) and the conv bias in front of BatchNorm has the ordinary gradient sum(dy)
        batch_stats = bool(ctx['training'])
        self._main_stream = torch.cuda.current_stream(dev) if dev.type == 'cuda' else None
        # every per-channel gradient of the step (BatchNorm gamma / beta sums, classifier bias) lives in ONE buffer: a
        # data-parallel run averages it with one collective instead of ~80 small ones (or a gather + scatter of them)
        pool_elems = roundup(self.head.cout, 64) if self.head is not None else 0
        for uc in ctx['units']:
            if uc.unit.main.has_bn or (uc.unit.res is not None and uc.unit.res.has_bn):
                pool_elems += 4 * acts[uc.out_index].CP
        # the two-launch BatchNorm-backward chain (FAST_BN_BWD): STAT_SLOTS zero rows per plain unit; the data gradients that form
        # the sums in their epilogue (w2l_conv1d_dgrad_bnreduce_ws) add onto the same rows (w2l_conv_stats_mode: thread-local,
        # this -- the autograd -- thread's launches)
        slot_need = 0
        if FAST_BN_BWD and not DETERMINISTIC_WGRAD and batch_stats and not self.precise and dev.type == 'cuda':
            slot_need = sum(STAT_SLOTS * 2 * acts[uc.out_index].CP for uc in ctx['units'] if uc.unit.main.has_bn and uc.unit.res is None)
        # the identically-zero gradients of conv biases in front of batch-statistics BatchNorm (_zeros)
        bias_need = 0
        if batch_stats:
            for u in self.units:
                for c in (u.main, u.res):
                    if c is not None and c.bias is not None and c.has_bn:
                        bias_need += roundup(c.cout, 64)
        # ONE zero-filled buffer (one fill launch on the caller's stream at the head of the backward pass, not three): the
        # per-channel gradient pool first -- a data-parallel run averages exactly that slice --, then the slot rows, then the
        # zero bias gradients
        zbuf = zeros(pool_elems + slot_need + bias_need, torch.float32, dev)
        small_pool = zbuf[:pool_elems]                           # rows of absent residual branches stay 0
        pool_off = 0
        self._slot_pool = [zbuf[pool_elems: pool_elems + slot_need], 0, {}] if slot_need else None
        self._zero_pool = [zbuf[pool_elems + slot_need:], 0] if bias_need else None
        lib.w2l_conv_stats_mode(STAT_SLOTS if self._slot_pool is not None else 0)
        lib.w2l_wgrad_deterministic(int(DETERMINISTIC_WGRAD))
        act_grads: List[List[tuple]] = [[] for _ in acts]
        if self.head is None:
            # open stack: the caller's gradient wrt the fp32 [N, C, T'] result becomes the (unpadded, fp32) gradient source
            last = acts[-1]
            _lib.poison('open stack')
            gp = torch.zeros(N, last.T, last.CP, dtype=torch.float32, device=dev)
            gp[:, :, :last.C] = g_out.float().transpose(1, 2)
            act_grads[len(acts) - 1].append((gp, 0, 0, PAD_ZERO, last.T))
        else:
            self._plan_wgrad_groups(ctx)
            pool_off = self._head_backward(ctx, g_out, small_pool, grads, act_grads)
        self._units_backward(ctx, act_grads, small_pool, pool_off, grads, batch_stats)
        for recs in list(self._wg_pending.values()):          # (a group whose members did not all arrive: never in a full backward)
            for r in recs:
                self._wgrad_single(r, grads)
        self._wg_pending.clear()
        self._wg_of = {}
        ctx['input_grad'] = self.input_grad(ctx, act_grads[0]) if ctx.get('want_dx') and act_grads[0] else None
        if self.flat_ready is not None:
            _py(self.flat_ready, small_pool)
        elif self.grad_ready is not None:
            _py(self.grad_ready, None, small_pool, small_pool)
        if self._side_used:
            if JOIN_EVENTS is not None:        # tools/stream_lag.py: which stream does the backward pass end on?
                em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                em.record(self._main_stream)
                es.record(self._side)
                JOIN_EVENTS.append((em, es))
            _lib.stream_wait_stream(self._main_stream, self._side)
            self._side_used = False
        self._main_stream = None
        self._held.clear()
        self._dyq.clear()
        self._zero_pool = None
        self._slot_pool = None
        lib.w2l_conv_stats_mode(0)
        _flush_tune_cache()
        if self.backward_done is not None:
            _py(self.backward_done)
        return [grads.get(id(p)) for p in self.parameters()]

    def _head_backward(self, ctx, g_out, small_pool, grads, act_grads) -> int:
        """(log_)softmax backward -> classifier weight / bias gradients and the gradient wrt the last activation;
        returns the number of pool elements it used"""
        precise = self.precise
        acts: List[Act] = ctx['acts']
        out = ctx['out']
        N, Th, L = out.shape
        dev = out.device
        st = stream_ptr
        pool_off = 0
        g_out = g_out.contiguous().float()
        glog = torch.empty_like(out)
        check(lib.w2l_log_softmax_bwd(ptr(g_out), ptr(out), N, Th, L, ctx['softmax_mode'], ptr(glog), st()),
              'w2l_log_softmax_bwd')
        head = self.head
        pk = pack_weights(head, precise)
        hh = roundup(Th, 64) - Th                    # shared-halo layout (include/w2l_hip.h): halo + N*(T+halo) rows
        dy_hi = torch.empty(hh + N * (Th + hh), pk.coutp, dtype=torch.bfloat16, device=dev)
        dy_lo = torch.empty_like(dy_hi) if precise else None
        colsum = small_pool[pool_off: pool_off + pk.coutp]
        pool_off += roundup(head.cout, 64)
        check(lib.w2l_pad_cast(ptr(glog), N, Th, L, pk.coutp, hh, ptr(dy_hi), ptr(dy_lo), ptr(colsum), st()),
              'w2l_pad_cast')
        last = acts[-1]
        self._wgrad(head, pk, dy_hi, dy_lo, hh, Th, last, grads)
        if head.bias is not None:
            grads[id(head.bias)] = colsum[: head.cout]          # travels with the pool
        act_grads[len(acts) - 1].append(self._dgrad(head, pk, dy_hi, dy_lo, hh, Th, last, self._producer(ctx, len(acts) - 1)))
        return pool_off

    def _producer(self, ctx, act_index: int):
        """the unit context whose BatchNorm-backward sums can be formed by the data gradient wrt activation ``act_index``
        (None: the spectrogram, a unit with a residual branch / without BatchNorm, the fp32 parity mode, or switched off)"""
        mode = FUSED_BN_REDUCE
        if mode in (False, '0') or self.precise or act_index < 1:
            return None
        uc = ctx['units'][act_index - 1]
        u = uc.unit
        if u.res is not None or not u.main.has_bn or uc.y is None or uc.y.dtype != torch.bfloat16:
            return None
        if mode == 'auto' and uc.y.shape[0] * uc.y.shape[1] >= FUSED_BN_REDUCE_MAX_ROWS:
            return None
        return (uc, ctx['training'])

    def _units_backward(self, ctx, act_grads, small_pool, pool_off, grads, batch_stats: bool):
        """the units in reverse: BatchNorm / activation backward -> dy -> weight gradient (side stream) -> data gradient"""
        precise = self.precise
        acts: List[Act] = ctx['acts']
        N = ctx['out'].shape[0]
        dev = ctx['out'].device
        st = stream_ptr
        amax_pool = (zeros((len(acts) + 1, 2, AMAX_SLOTS), torch.float32, dev)        # one fill per step
                     if self.fp8 else None)
        slot_pool = self._slot_pool    # zero rows the two-launch BatchNorm-backward chain adds its sums onto (backward())
        for uc in reversed(ctx['units']):
            u = uc.unit
            oi = uc.out_index
            srcs = act_grads[oi]
            if not srcs:
                raise RuntimeError('activation without a gradient source')
            if len(srcs) > 2:
                raise NotImplementedError('more than two consumers of one activation')
            a_out = acts[oi]
            Tout, coutp = uc.Tout, a_out.CP
            p = u.drop_p if (ctx['training'] and uc.mask is not None) else 0.0
            d = self._desc(uc, N, Tout, coutp, p, uc.lens_out)
            g1 = self._gsrc(srcs[0])
            g2 = self._gsrc(srcs[1]) if len(srcs) > 1 else None
            sums = None
            fold = fast = False
            if u.main.has_bn or (u.res is not None and u.res.has_bn):
                ncomp = 4 if u.res is not None else 2
                fused = [s_[5] for s_ in srcs if len(s_) > 5 and s_[5] is not None]
                fast = False
                if len(fused) == len(srcs) and ncomp == 2:
                    # every gradient source was a data-gradient convolution that formed the sums in its epilogue
                    fused = list({id(t): t for t in fused}.values())      # (two consumers may have added onto ONE set of slot rows)
                    partial = fused[0] if len(fused) == 1 else torch.cat(fused, 0)
                    nb = partial.shape[0]
                    # (on the step's slot rows: the dy pass with the finalize folded in takes them as they are)
                    fast = bool(slot_pool is not None and len(fused) == 1 and slot_pool[2].get(id(uc)) is not None and g2 is None
                                and lib.w2l_bn_bwd_fast_ok(C.byref(d), C.byref(g1), None))
                elif (FAST_BN_BWD and batch_stats and not precise and g2 is None and ncomp == 2 and slot_pool is not None
                        and lib.w2l_bn_bwd_fast_ok(C.byref(d), C.byref(g1), None)):
                    # the two-launch chain: sums added onto STAT_SLOTS zero rows of the step's pool, finalize folded into the dy pass
                    fast = True
                    nb = STAT_SLOTS
                    partial = slot_pool[0][slot_pool[1]: slot_pool[1] + nb * 2 * coutp].view(nb, 2, coutp)
                    slot_pool[1] += nb * 2 * coutp
                    check(lib.w2l_bn_act_bwd_reduce_slots(C.byref(d), C.byref(g1), ptr(partial), nb, st()), 'w2l_bn_act_bwd_reduce_slots')
                else:
                    nb = lib.w2l_bn_bwd_blocks(N, Tout, coutp)
                    partial = torch.empty(nb, ncomp, coutp, dtype=torch.float32, device=dev)
                    check(lib.w2l_bn_act_bwd_reduce(C.byref(d), C.byref(g1), C.byref(g2) if g2 else None, ptr(partial),
                                                    st()), 'w2l_bn_act_bwd_reduce')
                sums = small_pool[pool_off: pool_off + 4 * coutp].view(4, coutp)
                pool_off += 4 * coutp
                fold = (FOLD_BN_FINALIZE and batch_stats) or fast
                if not fold:
                    check(lib.w2l_bn_bwd_finalize(ptr(partial), nb, coutp, ncomp, ptr(sums), st()), 'w2l_bn_bwd_finalize')
                uc.keep.append((partial, sums))
            main, res = u.main, u.res
            need_dx_main = self._needs_grad(u.src, ctx)
            # wgrad walks each utterance in 64-row steps (the e4m3 kernel: 128-row steps) over zero rows
            tail = roundup(Tout, 128 if self.fp8 else 64) - Tout
            h1 = max((main.kernel - 1) * main.dilation, tail)
            dy_hi = torch.empty(h1 + N * (Tout + h1), coutp, dtype=torch.bfloat16, device=dev)
            dy_lo = torch.empty_like(dy_hi) if precise else None
            dy2_hi = dy2_lo = None
            h2 = 0
            if res is not None:
                h2 = max((res.kernel - 1) * res.dilation, tail)
                dy2_hi = torch.empty(h2 + N * (Tout + h2), coutp, dtype=torch.bfloat16, device=dev)
                dy2_lo = torch.empty_like(dy2_hi) if precise else None
            if not batch_stats:
                d = self._desc(uc, N, Tout, coutp, p, uc.lens_out, constant_stats=True)
            # fp8 mode: the kernel also leaves max |dy| (|dy2|) in device memory -- the scale of dy's e4m3 copy
            fp8_dgrad = self.fp8 and coutp % 128 == 0 and (FP8_DGRAD == '1' or (FP8_DGRAD == 'auto' and
                                                                                  N * Tout >= FP8_DGRAD_MIN_ROWS))
            fp8_wgrad = self.fp8 and coutp % 128 == 0 and (FP8_WGRAD == '1' or (FP8_WGRAD == 'auto' and
                                                                                  N * Tout >= FP8_DGRAD_MIN_ROWS))
            amax = amax_pool[oi] if fp8_dgrad or fp8_wgrad else None          # [2][AMAX_SLOTS]: dy, dy2
            amax_w = amax if fp8_wgrad else None
            amax_d = amax if fp8_dgrad else None
            if fold and fast:
                check(lib.w2l_bn_act_bwd_apply_slots(C.byref(d), C.byref(g1), ptr(partial), nb, ptr(sums), ptr(dy_hi), h1, ptr(amax),
                                                     st()), 'w2l_bn_act_bwd_apply_slots')
            elif fold:
                check(lib.w2l_bn_act_bwd_apply_fin(C.byref(d), C.byref(g1), C.byref(g2) if g2 else None, ptr(partial), nb,
                                                   ptr(sums), ptr(dy_hi), ptr(dy_lo), h1, ptr(dy2_hi), ptr(dy2_lo), h2, ptr(amax),
                                                   st()), 'w2l_bn_act_bwd_apply_fin')
            else:
                check(lib.w2l_bn_act_bwd_apply_amax(C.byref(d), C.byref(g1), C.byref(g2) if g2 else None, ptr(sums), ptr(dy_hi),
                                                    ptr(dy_lo), h1, ptr(dy2_hi), ptr(dy2_lo), h2, ptr(amax), st()),
                      'w2l_bn_act_bwd_apply')
            # release the consumed gradient buffers early
            act_grads[oi] = []
            # BN parameter gradients: d beta = sum g, d gamma = sum g * xhat
            if main.has_bn:
                grads[id(main.bn_bias)] = sums[0, : main.cout]
                grads[id(main.bn_weight)] = sums[1, : main.cout]
            if res is not None and res.has_bn:
                grads[id(res.bn_bias)] = sums[2, : res.cout]
                grads[id(res.bn_weight)] = sums[3, : res.cout]
            # main branch
            pkm = pack_weights(main, precise)
            src = acts[u.src] if u.dw is None else uc.mid
            # Which of the unit's two gradients is enqueued first decides what the weight gradient runs beside.  Weight
            # gradient first (rounds 1-4): its side-stream launch waits for dy only, so it competes with the data gradient of
            # the SAME unit -- two kernels that each fill the chip gain nothing from sharing it -- and is over when the next
            # unit's BatchNorm-backward chain (three short HBM-bound kernels + their stream boundaries, 60-90 us with the
            # matrix cores idle) begins.  Data gradient first (WGRAD_AFTER_DGRAD): the weight gradient's launch waits for the
            # data gradient too, i.e. it STARTS where that chain starts and runs beside it.
            late = WGRAD_AFTER_DGRAD and u.dw is None and need_dx_main
            if not late:
                self._wgrad(main, pkm, dy_hi, dy_lo, h1, Tout, src, grads, amax=None if amax_w is None else amax_w[0],
                            defer=self._defer_ok(ctx, uc, main))
            if main.bias is not None:
                if main.has_bn and batch_stats:      # sum(dy) == 0 identically under batch-statistics BatchNorm
                    grads[id(main.bias)] = self._zeros(main.cout, dev)      # zero on every rank: nothing to average
                else:
                    self._set(grads, main.bias, self._dy_colsum(dy_hi, dy_lo, h1, N, Tout, coutp, main.cout))
            if u.dw is not None:
                dmid = self._dgrad(main, pkm, dy_hi, dy_lo, h1, Tout, src,      # (wrt the depthwise output: no BatchNorm there)
                                   amax=None if amax_d is None else amax_d[0])
                gsrc = self._dw_backward(u.dw, dmid, acts[u.src], uc.mid, need_dx_main, grads)
                if gsrc is not None:
                    act_grads[u.src].append(gsrc)
            elif need_dx_main:
                act_grads[u.src].append(self._dgrad(main, pkm, dy_hi, dy_lo, h1, Tout, src, self._producer(ctx, u.src),
                                                    amax=None if amax_d is None else amax_d[0]))
            if late:
                self._wgrad(main, pkm, dy_hi, dy_lo, h1, Tout, src, grads, amax=None if amax_w is None else amax_w[0],
                            defer=self._defer_ok(ctx, uc, main))
            if res is not None:
                pkr = pack_weights(res, precise)
                rsrc = acts[u.res_src]
                self._wgrad(res, pkr, dy2_hi, dy2_lo, h2, Tout, rsrc, grads, amax=None if amax_w is None else amax_w[1],
                            defer=self._defer_ok(ctx, uc, res))
                if res.bias is not None:
                    if res.has_bn and batch_stats:
                        grads[id(res.bias)] = self._zeros(res.cout, dev)
                    else:
                        self._set(grads, res.bias, self._dy_colsum(dy2_hi, dy2_lo, h2, N, Tout, coutp, res.cout))
                if self._needs_grad(u.res_src, ctx):
                    act_grads[u.res_src].append(self._dgrad(res, pkr, dy2_hi, dy2_lo, h2, Tout, rsrc,
                                                            self._producer(ctx, u.res_src),
                                                            amax=None if amax_d is None else amax_d[1]))
    # ------------------------------------------------------------------ helpers
    def _needs_grad(self, act_index: int, ctx=None) -> bool:
        # the spectrogram needs no gradient in training (base_asr_models.py:78-85); computed only on request
        return act_index != 0 or bool(ctx and ctx.get('want_dx'))

    def input_grad(self, ctx, srcs) -> torch.Tensor:
        """gradient wrt the fp32 [N, C, T] spectrogram from the padded-coordinate gradient(s) of activation 0
        (cold path, torch ops: fold the reflected halo rows back, un-pad, back to channels-first)"""
        N, C0, T0 = ctx['x_shape']
        a0 = ctx['acts'][0]
        _lib.poison('spectrogram gradient')
        total = None
        for (g, pl, pr, mode, per, *_) in srcs:
            gv = g.view(N, per, a0.CP)[:, :pl + T0 + pr, :C0].float()
            core = gv[:, pl:pl + T0].clone()
            if mode == PAD_REFLECT:
                if pl:
                    core[:, 1:pl + 1] += gv[:, :pl].flip(1)
                if pr:
                    core[:, T0 - 1 - pr:T0 - 1] += gv[:, pl + T0:pl + T0 + pr].flip(1)
            total = core if total is None else total + core
        if a0.lens is not None:          # masked_fill on the input (jasper.py:116-119)
            t = torch.arange(T0, device=total.device)[None, :, None]
            total = total * (t < a0.lens.long()[:, None, None])
        return total.transpose(1, 2).contiguous()

    @staticmethod
    def _dy_colsum(dy_hi, dy_lo, halo, N, Tout, coutp, cout) -> torch.Tensor:
        """bias gradient sum_{n,t} dy of a conv that is not followed by batch-statistics BatchNorm (cold path, torch ops)"""
        _lib.poison('bias gradient of a convolution without BatchNorm')
        v = dy_hi[halo:].view(N, Tout + halo, coutp)[:, :Tout, :cout].float()
        if dy_lo is not None:
            v = v + dy_lo[halo:].view(N, Tout + halo, coutp)[:, :Tout, :cout].float()
        return v.sum((0, 1))

    def _zeros(self, n: int, dev) -> torch.Tensor:
        """an fp32 zero vector carved from one zero-filled buffer per backward (one fill launch instead of one per layer)"""
        need = roundup(n, 64)
        pool = self._zero_pool
        if pool is None or pool[0].device != dev or pool[1] + need > pool[0].numel():
            total = 0
            for u in self.units:
                for c in (u.main, u.res):
                    if c is not None and c.bias is not None and c.has_bn:
                        total += roundup(c.cout, 64)
            pool = [zeros(max(total, need), torch.float32, dev), 0]
            self._zero_pool = pool
        out = pool[0][pool[1]: pool[1] + n]
        pool[1] += need
        return out

    def _notify(self, param, grad, storage=None):
        if self.grad_ready is not None:
            _py(self.grad_ready, param, grad, storage)

    def _set(self, grads, param, grad, storage=None):
        grads[id(param)] = grad
        self._notify(param, grad, storage)

    def _gsrc(self, s) -> GradSrc:
        t, pl, pr, mode, rows = s[:5]
        g = GradSrc()
        g.dxp = t.data_ptr()
        g.f32 = int(t.dtype == torch.float32)
        g.pad_l, g.pad_r, g.pad_mode, g.rows = pl, pr, mode, rows
        return g

    def _dy_e4m3(self, dy_hi, amax):
        """dy's e4m3 copy and the device-side 1/scale (from the amax bn_act_bwd_apply left), made once per dy on the
        current stream and shared by the unit's weight and data gradients"""
        hit = self._dyq.get(id(dy_hi))
        if hit is None:
            dyq = torch.empty(dy_hi.shape, dtype=torch.uint8, device=dy_hi.device)
            inv = torch.empty(1, dtype=torch.float32, device=dy_hi.device)
            check(lib.w2l_quantize_e4m3_dyn(ptr(dy_hi), dy_hi.numel(), ptr(amax), ptr(dyq), ptr(inv), stream_ptr()),
                  'w2l_quantize_e4m3_dyn')
            hit = self._dyq[id(dy_hi)] = (dyq, inv, dy_hi)          # (dy_hi kept: its id must not be reused meanwhile)
        return hit[0], hit[1]

    def _defer_ok(self, ctx, uc: _UnitCtx, conv: ConvSpec) -> bool:
        """is this unit's weight gradient one of those held back until the next forward?"""
        k, opt = self.defer_wgrad, self.deferred
        if not k or opt is None or not ctx['training'] or self.precise:
            return False
        ui, n = uc.out_index - 1, len(self.units)
        if isinstance(k, int):                          # the top k units
            if ui < n - k:
                return False
        elif ui not in k and ui - n not in k:           # an explicit set of unit indices (negative: counted from the top)
            return False
        w = conv.weight
        # a gradient that autograd would ADD to an existing .grad (accumulation steps) is computed now; so is a second
        # gradient of a weight whose first one is still held back (two backward passes before one step(): the optimizer
        # adds the held one to .grad at step() -- one update with the sum, as torch.optim.SGD would make)
        if any(r['conv'].weight is w for r in self._deferred):
            return False
        return (w.is_cuda and w.grad is None and not torch.cuda.is_current_stream_capturing() and opt.accepts(w))

    def _defer_positions(self, recs):
        """WHERE in the forward pass each held-back weight gradient is launched (unit index; it is launched in front of
        that unit's convolution).  All of them at the start (round 4) run as one burst of long-lived full-chip kernels in
        the first ~3 ms of the forward -- the BatchNorm chains of the early, narrow layers are over-covered, those of the wide
        layers at the end (and the classifier / CTC tail) meet no matrix work at all -- and the forward's small dependent
        kernels queue behind whole rounds of their blocks.  W2L_DEFER_SPREAD: 'start' (the default: where a held-back gradient is
        launched inside the forward pass measured +-0.02 ms, profiles/r05_step_ab.txt) = round 4; 'la:K' = K units ahead of
        the layer whose update it carries (just in time, with margin); 'even' = evenly over the units in front of
        the first deferred layer.  A gradient is always launched strictly before its own layer's convolution."""
        mode = DEFER_SPREAD
        uis = []
        for r in recs:
            conv = r['conv']
            uis.append(next((i for i, u in enumerate(self.units) if u.main is conv or u.res is conv or u.dw is conv), 0))
        first = min(uis) if uis else 0
        for j, (r, ui) in enumerate(zip(recs, uis)):
            if mode == 'start':
                at = 0
            elif mode.startswith('la:'):
                at = ui - int(mode[3:])
            else:
                at = (j * max(first - 1, 0)) // max(len(recs), 1)
            r['at'] = max(0, min(at, ui - 1))

    def flush_deferred(self, pos=None):
        """launch the weight gradients held back by the last backward pass -- on the weight-gradient stream, in forward order,
        each one handed to the optimizer's fused update (which tags the weight's operand pack with an event the forward
        convolution of that layer waits for).  The forward pass calls it in front of every unit with that unit's index
        (``pos``): the gradients due there are launched (_defer_positions); without ``pos`` (optim.FusedSGD.join(): checkpoints,
        validation, the end of the forward) everything pending goes.
        If no optimizer step was taken since that backward (the caller only wanted gradients), the gradients are computed
        on the current stream and stored in ``param.grad`` instead."""
        recs, self._deferred = self._deferred, []
        if not recs:
            return
        recs.sort(key=lambda r: r['order'])
        opt = self.deferred
        if pos is not None:
            if any('at' not in r for r in recs):
                self._defer_positions([r for r in recs if opt is not None and opt.stepped(r['token'])])
            later = [r for r in recs if r.get('at', 0) > pos and opt is not None and opt.stepped(r['token'])]
            if later:
                self._deferred = later
                recs = [r for r in recs if not any(r is q for q in later)]
                if not recs:
                    return
        dev = recs[0]['dy_hi'].device
        main = torch.cuda.current_stream(dev)
        pending = []

        def sink(w, g, storage):
            pending.append((w, g, storage, self.grad_reduce_start(storage) if self.grad_reduce_start is not None else None))

        live = [r for r in recs if opt is not None and opt.stepped(r['token'])]
        stale = [r for r in recs if not (opt is not None and opt.stepped(r['token']))]
        self._materialize(stale)           # nobody stepped (and nobody called zero_grad): plain gradients, accumulated
        if not live:
            return
        fork = None
        if self.overlap_wgrad:
            if self._side is None or self._side.device != dev:
                self._side = _side_stream(dev, main)
            fork = (main, self._side)
        for r in live:
            self._wgrad_now(r['conv'], r['pk'], r['dy_hi'], r['dy_lo'], r['halo'], r['Tout'], r['src'], {}, fork=fork,
                            f8=r['f8'], sink=sink)
            if self.grad_reduce_start is None:           # one GPU: wgrad_i, update_i, wgrad_i+1, ... on one stream
                w, g, _, _ = pending.pop()
                with torch.cuda.stream(fork[1]) if fork else _nullctx():
                    opt.apply(w, g)
            if fork is not None:
                self._held.extend(t for t in (r['dy_hi'], r['dy_lo'], r['src'].hi, r['src'].lo) +
                                  ((r['src'].q,) + r['f8'] if r['f8'] else ()) if t is not None)
        if pending:                        # data parallel: every all-reduce is in flight before the first update waits for one
            with torch.cuda.stream(fork[1]) if fork else _nullctx():
                for w, g, _, work in pending:
                    work.finish()
                    opt.apply(w, g)
        if fork is not None:
            self._side_used = True

    def _materialize(self, recs):
        """held-back gradients computed NOW, on the caller's stream, and added to ``param.grad`` as autograd would have"""
        pending = []

        def sink(w, g, storage):
            pending.append((w, g, storage, self.grad_reduce_start(storage) if self.grad_reduce_start is not None else None))

        for r in recs:
            self._wgrad_now(r['conv'], r['pk'], r['dy_hi'], r['dy_lo'], r['halo'], r['Tout'], r['src'], {}, f8=r['f8'], sink=sink)
            w, g, _, work = pending.pop()
            if work is not None:
                work.finish()
            w.grad = g if w.grad is None else w.grad + g

    def drop_unstepped(self):
        """optimizer.zero_grad(): gradients held back by a backward pass that no step() followed are discarded with the
        rest (a skipped step must not resurface in the next one); those of a stepped batch stay -- they are that step's
        update, still to be applied"""
        opt = self.deferred
        self._deferred = [r for r in self._deferred if opt is not None and opt.stepped(r['token'])]

    def settle_before_step(self):
        """optimizer.step(): a held-back gradient whose weight ALSO has a ``.grad`` (a second backward pass ran before this
        step) joins it now, so that the weight gets ONE update with the summed gradient"""
        opt = self.deferred
        both = [r for r in self._deferred if r['conv'].weight.grad is not None and not (opt is not None and opt.stepped(r['token']))]
        if both:
            self._deferred = [r for r in self._deferred if not any(r is b for b in both)]
            self._materialize(both)

    def join_side(self):
        """the current stream waits for the weight-gradient stream (deferred gradients and their updates run there)"""
        if self._side is not None and self._side_used:
            _lib.stream_wait_stream(_lib.raw_stream(), self._side)

    def _wgrad(self, conv: ConvSpec, pk: _PackedW, dy_hi, dy_lo, halo, Tout, src: Act, grads, amax=None, defer=False):
        """dW, optionally on the side stream (ordered after everything enqueued so far on the current stream).

        No record_stream: dW is allocated (and zero-filled) on the main stream before the fork, and every tensor the
        side-stream kernel touches is kept alive in self._held until backward() joins the streams, so the caching
        allocator never has to poll cross-stream events (that polling stalled small-batch steps by 2-3x)."""
        f8 = None
        # (W2L_DETERMINISTIC=1: the e4m3 kernel sums its splits with fp32 atomics -- the bit-reproducible slab path is the
        # bf16 kernel's, so a deterministic run takes that one for the weight gradients in fp8 mode too)
        if (self.fp8 and not DETERMINISTIC_WGRAD and amax is not None and src.q is not None and conv.stride == 1 and pk.coutp % 128 == 0
                and src.CP == pk.cinp and pk.cinp % 128 == 0 and (min(conv.kernel, 2) - 1) * conv.dilation <= 32):
            f8 = self._dy_e4m3(dy_hi, amax)          # on the current (main) stream, before the fork
        if defer:
            self._deferred.append({'conv': conv, 'pk': pk, 'dy_hi': dy_hi, 'dy_lo': dy_lo, 'halo': halo, 'Tout': Tout, 'src': src,
                                   'f8': f8, 'order': len(self._deferred) * -1, 'token': self.deferred.token()})
            return
        grp = self._wg_of.get(id(conv)) if self._wg_of else None
        if grp is not None and f8 is None and dy_hi.is_cuda:
            recs = self._wg_pending.setdefault(grp[0], [])
            recs.append({'conv': conv, 'pk': pk, 'dy_hi': dy_hi, 'dy_lo': dy_lo, 'halo': halo, 'Tout': Tout, 'src': src})
            if len(recs) == grp[1]:
                del self._wg_pending[grp[0]]
                self._wgrad_group_now(recs, grads)
            return
        if not self.overlap_wgrad or not dy_hi.is_cuda:
            return self._wgrad_now(conv, pk, dy_hi, dy_lo, halo, Tout, src, grads, f8=f8)
        main = self._main_stream or torch.cuda.current_stream(dy_hi.device)
        if self._side is None or self._side.device != dy_hi.device:
            # ONE stream per device for the life of the process (engines are rebuilt per forward).  HIP multiplexes streams
            # onto 4 hardware queues round-robin: a fresh pool stream per step lands on the main stream's queue every fourth
            # step and that step loses the overlap (14.8 instead of 13.8 ms; Jasper 23.4 instead of 19.7).  Default priority:
            # streams created with hipStreamCreateWithPriority (lowest OR highest) made the whole step 30 % slower.
            self._side = _side_stream(dy_hi.device, main)
        side = self._side
        self._wgrad_now(conv, pk, dy_hi, dy_lo, halo, Tout, src, grads, fork=(main, side), f8=f8)
        self._held.extend(t for t in (dy_hi, dy_lo, src.hi, src.lo) + ((src.q,) + f8 if f8 else ()) if t is not None)
        self._side_used = True

    # ------------------------------------------------------------------ grouped weight gradients
    def _plan_wgrad_groups(self, ctx):
        """which weight gradients of this backward pass share a launch (wgrad_groups.plan over the convolutions in the order
        their dy appears: classifier, then the units top down, main branch before residual branch)"""
        self._wg_of, self._wg_pending = {}, {}
        setting = WG.setting()
        if self.precise or self.fp8 or setting.strip().lower() in ('0', 'off', 'none') or not ctx['training']:
            return
        # data parallel: a group's gradients reach the reducer when the WHOLE group's kernel is done -- the all-reduce of its
        # first members starts up to two layers late.  Measured with one rank and the RCCL path forced (round 6, same box,
        # communicator created first as in a real run): 13.10 / 13.14 ms with groups against 13.36 / 13.49 without -- the groups'
        # gain outweighs the later start when nothing has to cross a link; W2L_WGRAD_GROUPS_DP=0 plans no groups while a reducer
        # is attached (the switch to try on a real node if the exchange turns out exposed)
        if self.grad_ready is not None and os.environ.get('W2L_WGRAD_GROUPS_DP', '1') == '0' and setting.strip().lower() in ('auto', '1', ''):
            return
        acts: List[Act] = ctx['acts']
        N = ctx['out'].shape[0]
        convs, seq = [], []

        def add(conv, src, Tout, uc):
            ok = (conv.stride == 1 and conv.weight.is_cuda and (uc is None or not self._defer_ok(ctx, uc, conv)))
            pk = pack_weights(conv, False) if ok else None
            ok = ok and src.CP == pk.cinp and src.lo is None
            convs.append(conv)
            seq.append((pk.cinp, pk.coutp, conv.kernel, conv.dilation, (N, Tout)) if ok else None)

        if self.head is not None:
            add(self.head, acts[-1], ctx['out'].shape[1], None)
        for uc in reversed(ctx['units']):
            u = uc.unit
            add(u.main, acts[u.src] if u.dw is None else uc.mid, uc.Tout, uc)
            if u.res is not None:
                add(u.res, acts[u.res_src], uc.Tout, uc)
        key = (tuple(seq), setting, WGROUP_MAX)
        groups = _wgroup_plans.get(key)
        if groups is None:
            groups = WG.parse_override(setting, len(seq))
            if groups is None:
                groups = WG.plan(seq, max_group=WGROUP_MAX)
            groups = [g for g in groups if all(seq[i] is not None for i in g)
                      and len({(seq[i][3], seq[i][4]) for i in g}) == 1 and len(g) <= WG.MAX_GROUP]
            _wgroup_plans[key] = groups
        for gi, g in enumerate(groups):
            for i in g:
                self._wg_of[id(convs[i])] = (gi, len(g))
        self._wg_seen = dict(self._wg_of)          # (what the last backward pass planned: read by tests)

    def _wgrad_single(self, r, grads, sink=None):
        """one member of a group through the ordinary path (its own measured plan)"""
        conv, dy_hi = r['conv'], r['dy_hi']
        if not self.overlap_wgrad or sink is not None:
            return self._wgrad_now(conv, r['pk'], dy_hi, r['dy_lo'], r['halo'], r['Tout'], r['src'], grads, sink=sink)
        main = self._main_stream or torch.cuda.current_stream(dy_hi.device)
        if self._side is None or self._side.device != dy_hi.device:
            self._side = _side_stream(dy_hi.device, main)
        self._wgrad_now(conv, r['pk'], dy_hi, r['dy_lo'], r['halo'], r['Tout'], r['src'], grads, fork=(main, self._side))
        self._held.extend(t for t in (dy_hi, r['dy_lo'], r['src'].hi, r['src'].lo) if t is not None)
        self._side_used = True

    def _wgrad_group_now(self, recs, grads):
        """the weight gradients of ``recs`` (same N, Tout, dilation; stride 1) in ONE launch (w2l_conv1d_wgrad_group) -- or one
        by one, if that measured faster for this group (decided once per group signature, during warm-up)"""
        dev = recs[0]['dy_hi'].device
        N, Tout, dil = recs[0]['src'].N, recs[0]['Tout'], recs[0]['conv'].dilation
        key = (tuple((r['pk'].cinp, r['pk'].coutp, r['conv'].kernel) for r in recs), N, Tout, dil)
        form = _wgroup_forms.get(key)
        if form is None:
            form = self._wgrad_group_measure(recs, key) if AUTOTUNE and not torch.cuda.is_current_stream_capturing() else \
                WG.best_cost([k for k in key[0]], dil, False)[1]
            _wgroup_forms[key] = form
        if form < 0:
            for r in recs:
                self._wgrad_single(r, grads)
            return
        main = self._main_stream or torch.cuda.current_stream(dev)
        fork = None
        if self.overlap_wgrad:
            if self._side is None or self._side.device != dev:
                self._side = _side_stream(dev, main)
            fork = (main, self._side)
        dws = self._wgrad_group_launch(recs, form, fork)
        # the gradients are handed over ON THE STREAM THAT WROTE THEM: a data-parallel reducer (distributed.GradReducer.on_grad)
        # orders its all-reduce behind an event it records on the current stream -- on the caller's stream that event would
        # say nothing about the group kernel, and the collective could read (and overwrite in place) dW while it is written
        with torch.cuda.stream(fork[1]) if fork is not None else _nullctx():
            for r, dw in zip(recs, dws):
                w = r['conv'].weight
                cout, cin, kw = w.shape
                g = dw.permute(1, 2, 0)                     # logical [CoutP, CinP, Kw]
                if not (r['pk'].coutp == cout and r['pk'].cinp == cin):
                    g = g[:cout, :cin, :]
                self._set(grads, w, g, storage=dw)

    def _wgrad_group_launch(self, recs, form, fork):
        dev = recs[0]['dy_hi'].device
        N, Tout, dil = recs[0]['src'].N, recs[0]['Tout'], recs[0]['conv'].dilation
        items = (_lib.WgradItem * len(recs))()
        dws = []
        for it, r in zip(items, recs):
            conv, pk, src, halo = r['conv'], r['pk'], r['src'], r['halo']
            row_off = src.pad_l - conv.pad_l
            # (a group launch stores whole tiles: a zero-filled buffer optim.FusedSGD left on the weight is of no use here --
            # dropped, so that it does not stay alive beside the gradient)
            conv.weight.__dict__.pop('_w2l_dw_zeroed', None)
            dw = torch.empty(conv.kernel, pk.coutp, pk.cinp, dtype=torch.float32, device=dev)      # (on the caller's stream)
            dws.append(dw)
            it.dy = r['dy_hi'].data_ptr() + halo * pk.coutp * 2
            it.dy_bstride = (Tout + halo) * pk.coutp
            it.xp = src.hi.data_ptr() + row_off * src.CP * 2
            it.x_bstride = src.rows * src.CP
            it.x_rows_total = N * src.rows - row_off
            it.dw = dw.data_ptr()
            it.Cin, it.Cout, it.Kw = pk.cinp, pk.coutp, conv.kernel
        flops = sum(2.0 * N * Tout * r['pk'].coutp * r['pk'].cinp * r['conv'].kernel for r in recs)
        if fork is not None:
            _lib.stream_wait_stream(fork[1], fork[0])
            self._held.extend(dws)
            self._held.extend(t for r in recs for t in (r['dy_hi'], r['src'].hi) if t is not None)
            self._side_used = True
        with torch.cuda.stream(fork[1]) if fork is not None else _nullctx():
            with _timed('conv_wgrad3_kernel' if form & 16 else 'conv_wgrad_kernel', flops):
                check(lib.w2l_conv1d_wgrad_group(items, len(recs), N, Tout, dil, form, stream_ptr()), 'w2l_conv1d_wgrad_group')
        return dws

    def _wgrad_group_measure(self, recs, key) -> int:
        """SYNCHRONISING, warm-up only: the group in every block form against its members one by one with their own measured
        plans (zero fills included); returns the fastest form, or -1 for 'one by one'"""
        dev = recs[0]['dy_hi'].device
        main = torch.cuda.current_stream(dev)
        if self._side is not None:
            main.wait_stream(self._side)

        def timed(fn, reps=3):
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(main)
            for _ in range(reps):
                fn()
            e.record(main)
            e.synchronize()
            return s.elapsed_time(e) / reps

        def singles():
            for r in recs:
                self._wgrad_single(r, {}, sink=lambda *a: None)
        best = (timed(singles), -1)
        report = [f'one by one {best[0]:.3f}']
        for form in WG.forms_for(key[3]):
            try:
                t = timed(lambda: self._wgrad_group_launch(recs, form, None))
            except RuntimeError:
                continue
            report.append(f'form {form} {t:.3f}')
            best = min(best, (t, form))
        if os.environ.get('W2L_WGRAD_GROUPS_VERBOSE'):
            print(f'[w2l] wgrad group {key[0]} N={key[1]} T={key[2]} d={key[3]}: ' + ', '.join(report) + f' ms -> {best[1]}', flush=True)
        _tune_state['dirty'] = True
        return best[1]

    def _wgrad_now(self, conv: ConvSpec, pk: _PackedW, dy_hi, dy_lo, halo, Tout, src: Act, grads, fork=None, f8=None, sink=None):
        """dW through w2l_conv1d_wgrad, written in the parameter's own physical layout when possible.
        fork=(main, side): allocate on main, launch on side after an event recorded on main."""
        w = conv.weight
        cout, cin, kw = w.shape
        dev = w.device
        N = src.N
        direct = (pk.coutp == cout and pk.cinp == cin)
        row_off = src.pad_l - conv.pad_l
        x_bstride = src.rows * src.CP
        x_rows_total = N * src.rows - row_off
        dy_bstride = (Tout + halo) * pk.coutp          # shared-halo layout: utterance n starts at row halo + n*(Tout+halo)
        ws = _wgrad_workspace(dev, pk.cinp, pk.coutp, kw) if f8 is None and (DETERMINISTIC_WGRAD or DEALT_WGRAD) else None
        ws_bytes = ws.numel() if ws is not None else 0
        if f8 is not None:
            if AUTOTUNE:
                key = ('wgrad_fp8', N, pk.cinp, pk.coutp, Tout, kw, conv.dilation, dev.index)
                if key not in _tuned_shapes:
                    _tuned_shapes.add(key)
                    _tune_state['dirty'] = True
                    scratch = torch.empty(kw, pk.coutp, pk.cinp, dtype=torch.float32, device=dev)
                    check(lib.w2l_conv1d_wgrad_fp8_tune(C.c_void_p(f8[0].data_ptr() + halo * pk.coutp), dy_bstride,
                                                        C.c_void_p(src.q.data_ptr() + row_off * src.CP), x_bstride, x_rows_total,
                                                        ptr(scratch), N, pk.cinp, pk.coutp, Tout, kw, conv.dilation, TUNE_REPS,
                                                        stream_ptr()), 'w2l_conv1d_wgrad_fp8_tune')
        elif AUTOTUNE and not self.precise:
            key = ('wgrad', N, pk.cinp, pk.coutp, Tout, kw, conv.stride, conv.dilation, dev.index)
            if key not in _tuned_shapes:       # once per shape and device, during the first (warm-up) step
                _tuned_shapes.add(key)
                _tune_state['dirty'] = True
                scratch = torch.empty(kw, pk.coutp, pk.cinp, dtype=torch.float32, device=dev)
                if self._side is not None:         # the workspace is shared with gradients still running on the side stream
                    torch.cuda.current_stream(dev).wait_stream(self._side)
                check(lib.w2l_conv1d_wgrad_tune_x(C.c_void_p(dy_hi.data_ptr() + halo * pk.coutp * 2), dy_bstride,
                                                  C.c_void_p(src.hi.data_ptr() + row_off * src.CP * 2), x_bstride, x_rows_total,
                                                  ptr(scratch), N, pk.cinp, pk.coutp, Tout, kw, conv.stride, conv.dilation, TUNE_REPS,
                                                  ptr(ws), ws_bytes, 0 if DETERMINISTIC_WGRAD else 1, stream_ptr()),
                      'w2l_conv1d_wgrad_tune_x')
        # with a workspace, split reductions end in plain stores by the last block of a tile: no zero fill, no atomics
        if f8 is not None:
            need_zero = bool(lib.w2l_wgrad_fp8_needs_zero(N, pk.cinp, pk.coutp, Tout, kw))
        else:
            need_zero = bool(lib.w2l_wgrad_needs_zero_x(N, pk.cinp, pk.coutp, Tout, kw, conv.stride, conv.dilation, ws_bytes)) or self.precise
        # optim.FusedSGD leaves last step's gradient buffer zero-filled on the parameter: take it as this step's dW (only
        # when zero_grad(set_to_none=True) dropped p.grad -- otherwise autograd is about to ADD into that very tensor)
        recycled = w.__dict__.pop('_w2l_dw_zeroed', None)
        if (recycled is not None and need_zero and w.grad is None and recycled.device == dev and recycled.is_contiguous()
                and tuple(recycled.shape) == (kw, pk.coutp, pk.cinp)):
            if fork is not None:
                _lib.stream_wait_stream(fork[1], fork[0])
                self._held.append(recycled)
                with torch.cuda.stream(fork[1]):
                    return self._wgrad_launch(conv, pk, recycled, dy_hi, dy_lo, halo, Tout, src, grads, x_bstride,
                                              x_rows_total, dy_bstride, row_off, direct, ws, f8, sink)
            return self._wgrad_launch(conv, pk, recycled, dy_hi, dy_lo, halo, Tout, src, grads, x_bstride, x_rows_total,
                                      dy_bstride, row_off, direct, ws, f8, sink)
        if fork is not None:
            # allocated on the main stream (the caching allocator then owns it there), zero-filled on the side stream:
            # the fill of a split-K gradient is as far off the critical path as the kernel that accumulates into it
            dw = torch.empty(kw, pk.coutp, pk.cinp, dtype=torch.float32, device=dev)
            _lib.stream_wait_stream(fork[1], fork[0])
            self._held.append(dw)
            with torch.cuda.stream(fork[1]):
                if need_zero:
                    zero_(dw)
                return self._wgrad_launch(conv, pk, dw, dy_hi, dy_lo, halo, Tout, src, grads, x_bstride, x_rows_total,
                                          dy_bstride, row_off, direct, ws, f8, sink)
        dw = (zeros if need_zero else (lambda shape, dtype, device: torch.empty(shape, dtype=dtype, device=device)))(
            (kw, pk.coutp, pk.cinp), torch.float32, dev)
        return self._wgrad_launch(conv, pk, dw, dy_hi, dy_lo, halo, Tout, src, grads, x_bstride, x_rows_total, dy_bstride,
                                  row_off, direct, ws, f8, sink)

    def _wgrad_launch(self, conv, pk, dw, dy_hi, dy_lo, halo, Tout, src, grads, x_bstride, x_rows_total, dy_bstride, row_off,
                      direct, ws, f8=None, sink=None):
        w = conv.weight
        cout, cin, kw = w.shape
        N = src.N
        st = stream_ptr()

        def run(dy, x, acc):
            check(lib.w2l_conv1d_wgrad_ws(C.c_void_p(dy.data_ptr() + halo * pk.coutp * 2), dy_bstride,
                                          C.c_void_p(x.data_ptr() + row_off * src.CP * 2), x_bstride, x_rows_total,
                                          ptr(dw), N, pk.cinp, pk.coutp, Tout, kw, conv.stride, conv.dilation, acc, ptr(ws),
                                          ws.numel() if ws is not None else 0, st), 'w2l_conv1d_wgrad_ws')

        if f8 is not None:
            # fp8 mode: dy's e4m3 copy x the input's e4m3 copy (the operand of the forward convolution), fp32 result
            with _timed('conv_wgrad_fp8_kernel', 2.0 * N * Tout * cout * cin * kw):
                check(lib.w2l_conv1d_wgrad_fp8(C.c_void_p(f8[0].data_ptr() + halo * pk.coutp), dy_bstride,
                                               C.c_void_p(src.q.data_ptr() + row_off * src.CP), x_bstride, x_rows_total, ptr(dw), N,
                                               pk.cinp, pk.coutp, Tout, kw, conv.dilation, 1.0 / src.q_scale, ptr(f8[1]), 0, st),
                      'w2l_conv1d_wgrad_fp8')
        elif not self.precise:
            # (which kernel family: the three-tap AGPR code object -- plan order bit 4 on a stride-1, dilation <= 4 layer -- or the
            # two-tap kernels; bench.py reports the two populations side by side)
            label = 'conv_wgrad_kernel'
            if KERNEL_TIMER is not None and conv.stride == 1 and conv.dilation <= 4 and kw >= 3 and \
                    lib.w2l_wgrad_plan(N, pk.cinp, pk.coutp, Tout, kw) & 16:
                label = 'conv_wgrad3_kernel'
            with _timed(label, 2.0 * N * Tout * cout * cin * kw):
                run(dy_hi, src.hi, 0)
        else:
            run(dy_hi, src.hi, 1)
            run(dy_hi, src.lo, 1)
            run(dy_lo, src.hi, 1)
        g = dw.permute(1, 2, 0)                     # logical [CoutP, CinP, Kw]
        if not direct:
            g = g[:cout, :cin, :]
        if sink is not None:                        # a deferred gradient: reduced / applied by flush_deferred, not by autograd
            return sink(w, g, dw)
        self._set(grads, w, g, storage=dw)

    def _dgrad_fused(self, conv: ConvSpec, pk: _PackedW, dy_hi, halo, hb, per, flat_rows, total, dxp, src: Act, producer,
                     flops) -> torch.Tensor:
        """data gradient + the BatchNorm-backward sums of the layer that produced ``src`` (w2l_conv1d_dgrad_bnreduce_ws);
        returns partial [tiles][2][C]"""
        uc, training = producer
        dev = dxp.device
        p = uc.unit.drop_p if (training and uc.mask is not None) else 0.0
        d = self._desc(uc, src.N, src.T, src.CP, p, uc.lens_out)
        pool = self._slot_pool
        slots = pool is not None and uc.unit.res is None
        if slots:                           # (w2l_conv_stats_mode(STAT_SLOTS) is in force: the epilogue ADDS onto these zero rows)
            partial = pool[2].get(id(uc))   # a second consumer of the same activation (a residual branch) adds onto the same rows
            if partial is None:
                n = STAT_SLOTS * 2 * src.CP
                partial = pool[0][pool[1]: pool[1] + n].view(STAT_SLOTS, 2, src.CP)
                pool[1] += n
                pool[2][id(uc)] = partial
        else:
            if pool is not None:
                lib.w2l_conv_stats_mode(0)
            tiles = lib.w2l_conv_stat_tiles(1, flat_rows)
            partial = torch.empty(tiles, 2, src.CP, dtype=torch.float32, device=dev)
        row_off = halo - hb
        dy_ptr = C.c_void_p(dy_hi.data_ptr() + row_off * pk.coutp * 2)
        rows_total = total - row_off
        ws = _splitk_workspace(dev, 1, pk.cinp, flat_rows)
        st = stream_ptr()
        tail = (C.byref(d), conv.pad_l, conv.pad_r, conv.pad_mode, per, pk.coutp, flat_rows, conv.kernel, conv.dilation)
        args = (dy_ptr, rows_total, ptr(pk.dgr_hi), ptr(dxp), ptr(partial)) + tail
        if AUTOTUNE:
            key = ('dgrad+bn', pk.coutp, pk.cinp, flat_rows, conv.kernel, conv.dilation, dev.index)
            if key not in _tuned_shapes:
                _tuned_shapes.add(key)
                _tune_state['dirty'] = True
                # the measuring launches ADD to slot rows: they get rows of their own -- the step's rows may already hold the
                # sums another consumer of the same activation (a residual branch) has added
                scratch = torch.zeros_like(partial) if slots else partial
                check(lib.w2l_conv1d_dgrad_bnreduce_tune_ws(dy_ptr, rows_total, ptr(pk.dgr_hi), ptr(dxp), ptr(scratch), *tail,
                                                            TUNE_REPS, ptr(ws), ws.numel(), st),
                      'w2l_conv1d_dgrad_bnreduce_tune_ws')
        with _timed('conv_igemm_kernel/dgrad+bnreduce', flops):
            check(lib.w2l_conv1d_dgrad_bnreduce_ws(*args, ptr(ws), ws.numel(), st), 'w2l_conv1d_dgrad_bnreduce_ws')
        if pool is not None and not slots:
            lib.w2l_conv_stats_mode(STAT_SLOTS)
        return partial

    def _dgrad(self, conv: ConvSpec, pk: _PackedW, dy_hi, dy_lo, halo, Tout, src: Act, producer=None, amax=None):
        """dXpad (gradient wrt the conv's padded input) through the same implicit-GEMM kernel, run over
        the shared-halo dy buffer as ONE sequence of N*(Tout+halo) rows: tiles never straddle a partially
        filled per-utterance remainder.  Row v of utterance n lands at output row n*(Tout+halo) + v."""
        N = src.N
        Tp = src.T + conv.pad_l + conv.pad_r
        hb = (conv.kernel - 1) * conv.dilation
        if conv.stride != 1:
            # cold path (only the spectrogram's gradient needs it): dy is zero-stuffed to stride 1 -- dy_up[t*s] = dy[t]
            # -- and the stride-1 data gradient runs on that; rows past (Tout-1)*s + hb of the padded input get none
            s_ = conv.stride
            Tup = (Tout - 1) * s_ + 1
            hup = max(hb, roundup(Tup, 64) - Tup)

            def stuff(t):
                if t is None:
                    return None
                _lib.poison('strided data gradient')
                up = torch.zeros(hup + N * (Tup + hup), pk.coutp, dtype=t.dtype, device=t.device)
                up[hup:].view(N, Tup + hup, pk.coutp)[:, 0:Tup:s_] = t[halo:].view(N, Tout + halo, pk.coutp)[:, :Tout]
                return up

            dy_hi, dy_lo, halo, Tout = stuff(dy_hi), stuff(dy_lo), hup, Tup
        assert Tout + hb <= Tp and halo >= hb
        dev = dy_hi.device
        per = Tout + halo
        flat_rows = N * per
        dxp = torch.empty(flat_rows, pk.cinp, dtype=torch.float32 if self.precise else torch.bfloat16, device=dev)
        total = dy_hi.shape[0]
        flops = 2.0 * N * Tout * conv.cout * conv.cin * conv.kernel
        if self.fp8 and amax is not None and conv.stride == 1 and pk.coutp % 128 == 0 and per >= Tp:
            # fp8 mode: dy's e4m3 copy (scale from the device-side amax bn_act_bwd_apply left) x the e4m3 flipped-tap weights
            dyq, inv = self._dy_e4m3(dy_hi, amax)
            wq, w_scale = _fp8_weights(conv, pk, dgrad=True)
            row_off = halo - hb
            xq = C.c_void_p(dyq.data_ptr() + row_off * pk.coutp)
            rows_total = total - row_off
            st_ = stream_ptr()
            if AUTOTUNE:
                key = ('fp8', 1, pk.coutp, pk.cinp, flat_rows, conv.kernel, conv.dilation, False, dev.index)
                if key not in _tuned_shapes:
                    _tuned_shapes.add(key)
                    _tune_state['dirty'] = True
                    check(lib.w2l_conv1d_igemm_fp8_tune(xq, rows_total * pk.coutp, rows_total, ptr(wq), ptr(dxp), 0, None, None, 1,
                                                        pk.coutp, pk.cinp, flat_rows, conv.kernel, conv.dilation, TUNE_REPS, st_),
                          'w2l_conv1d_igemm_fp8_tune')
            with _timed('conv_igemm_fp8_kernel', flops):
                check(lib.w2l_conv1d_igemm_fp8(xq, rows_total * pk.coutp, rows_total, ptr(wq), ptr(dxp), 0, 1.0 / w_scale, ptr(inv),
                                               None, None, 1, pk.coutp, pk.cinp, flat_rows, conv.kernel, conv.dilation, st_),
                      'w2l_conv1d_igemm_fp8')
            return (dxp, conv.pad_l, conv.pad_r, conv.pad_mode, per)
        if producer is not None and conv.stride == 1 and per >= Tp and src.CP == pk.cinp:
            partial = self._dgrad_fused(conv, pk, dy_hi, halo, hb, per, flat_rows, total, dxp, src, producer, flops)
            return (dxp, conv.pad_l, conv.pad_r, conv.pad_mode, per, partial)
        dyact = Act(dy_hi, dy_lo, 1, total, pk.coutp, pk.coutp, 0, 0, PAD_ZERO)
        _igemm(dyact, halo - hb, pk.dgr_hi, pk.dgr_lo, dxp, None, None, pk.coutp, pk.cinp, flat_rows, conv.kernel, 1,
               conv.dilation, self.precise, alg_flops=flops)
        if per < Tp:                      # strided case: the last (Tp - Tup - hb) padded rows receive no gradient
            _lib.poison('strided data gradient')
            full = torch.zeros(N, Tp, pk.cinp, dtype=dxp.dtype, device=dev)
            full[:, :per] = dxp.view(N, per, pk.cinp)
            return (full, conv.pad_l, conv.pad_r, conv.pad_mode, Tp)
        return (dxp, conv.pad_l, conv.pad_r, conv.pad_mode, per)
