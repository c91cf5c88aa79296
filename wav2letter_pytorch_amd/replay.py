"""Recorded launch lists: the training step replayed through ONE C-ABI call per phase (w2l_replay, include/w2l_hip.h).

The eager step engine (engine.py) decides in Python, for every launch of every step, what it already decided the step before:
200 (Wav2Letter) to 500 (Jasper 10x5) Python -> ctypes transitions, tensor allocations and stream / event calls per step --
4.4-5.0 ms and 12.3 ms of host time (DESIGN Appendix C).  That is invisible while the GPU needs 12.5 ms per step and IS the
step time everywhere else: small batches, the shipped default ``mid_layers: 1`` (configuration/model/wav2letter.yaml:3), Jasper
in fp8 mode.  Once a step shape is warm (every kernel plan measured, every persistent buffer in place) the engine therefore
RECORDS the step while running it -- which entry point, which argument values, which stream, which event records / waits
between the streams -- and from then on replays it:

  phase F   forward pass of the stack (everything engine.forward enqueues)
  phase B   backward pass (engine.backward: BatchNorm / activation backward, data and weight gradients, the join of the
            weight-gradient stream)
  phase O   optimizer step (optim.FusedSGD.step: the fused conv-weight updates on the optimizer's stream, the small parameters
            in one launch)
  phase X   the weight gradients + fused updates a backward pass held back for the next forward pass (flush_deferred)

each through one w2l_replay call: a C loop over the same extern "C" entry points, every launch on the stream it was recorded
on.  This is NOT a hipGraph: nothing is captured, joined or instantiated; the side streams, the optimizer's updates streaming
under the next forward pass and the held-back weight gradients overlap exactly as in the eager step (a graph is one unit: the
next one waits for all of it -- measured 4-12 % slower at N >= 8, DESIGN Appendix C).

What makes a step replayable
  * Static memory.  A record owns torch.cuda.MemPools: every tensor the recorded step allocates (activations, gradients,
    masks, workspaces of the step) comes from them and nothing else ever does, so the recorded addresses stay valid; blocks the
    step freed and reused (dy buffers) alias exactly as they did in the eager step, under the same stream order.  One pool for
    forward + backward (one eager timeline), one each for the optimizer phase and for phase X, a fresh one for every recording
    attempt: whatever is allocated LATER -- a persistent table built by a dropped recording, the held-back gradients' dW --
    can never land in a block the forward / backward record still writes as a transient buffer.
  * Two record sets per step shape, used alternately.  The held-back weight gradients of step i read step i's activations
    while the forward pass of step i + 1 runs: with one set of buffers step i + 1 would overwrite what they read.  Eager steps
    get fresh buffers from the allocator; replayed steps alternate between set A and set B.
  * Stream order through entry points (w2l_event_record, w2l_stream_wait_event, w2l_stream_wait_stream) and one persistent
    event per conv weight (engine.weight_event): a recorded forward waits for the same event handles whichever step -- eager
    or replayed, set A or B -- updated the weights before it.
  * No torch ops between the launches of a phase: fills, padded per-channel vectors, BatchNorm counters, the small-parameter
    SGD update and the dropout step counter are entry points too (csrc/replay.hip).  Code paths that still need torch ops or
    a host synchronisation (measuring launches, fp8 scale upkeep, gradients of the spectrogram, ...) ``poison`` a recording:
    that step simply stays eager and the next one tries again.
  * The batch: the spectrograms are copied into the set's static input buffer (skipped when the caller hands over the very
    tensor the set was recorded with), Jasper's length chain (jasper.py:109-121) is re-evaluated on the host and uploaded into
    the set's static length table, the gradient wrt the log-probabilities is copied into the set's static buffer.
  * Dropout: the Philox offset of nn.Dropout (wav2letter.py:38,44) is (unit index) + a step counter in device memory that a
    recorded w2l_counter_add bumps -- fresh masks every replay (the mechanism of graph.GraphedTrainStep).

Not replayed (the step stays eager, silently): evaluation-mode forwards, the fp32 parity mode's debug hooks, a spectrogram that
needs a gradient, device-side lengths, data-parallel runs (torch.distributed collectives are not entry points), hipGraph
capture, any kernel timer / launch trace, ``W2L_REPLAY=0``.

Semantics that differ from the eager step: the tensor ``forward`` returns is the set's static output buffer -- it is
overwritten when the same set runs again, two steps later (read it, or copy it, before that); ``p.grad`` of every parameter is
assigned by the backward pass itself (static gradient buffers), not accumulated by autograd -- a backward pass that finds
``p.grad`` already set falls back to the eager path for that step.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr

ENABLED = os.environ.get('W2L_REPLAY', '1') != '0'
WARM_STEPS = int(os.environ.get('W2L_REPLAY_WARM', '2'))          # eager steps of a shape before it is recorded
MAX_GROUPS = int(os.environ.get('W2L_REPLAY_MAX_SHAPES', '8'))    # step shapes kept recorded per engine (least recently used out)
MAX_FAILURES = 3
FP8_EAGER_EVERY = 256           # fp8 mode: replayed steps between two eager stretches (the e4m3 weight-scale upkeep lives there)
FP8_EAGER_STEPS = 10
VERBOSE = os.environ.get('W2L_REPLAY_VERBOSE', '0') == '1'
LENS_RING = 4
# reasons for dropping a recording that pass by themselves (one-time set-up, the periodic e4m3 scale upkeep, plans still being
# measured): the shape is tried again after its warm steps, without counting towards MAX_FAILURES
TRANSIENT = ('fp8 weight scale', 'kernel plans were measured', 'measuring launches', 'workspace grown', 'created', 'table built',
             'momentum buffer')

# the record set whose backward pass ran last on this process (a weak reference to its replayer + the set): how
# optim.FusedSGD.step finds out that the gradients it is about to consume are a record's static buffers
_last_backward = [None]
STATS = {'recorded': 0, 'replayed_F': 0, 'replayed_B': 0, 'replayed_O': 0, 'replayed_X': 0, 'poisoned': []}


def _say(msg):
    if VERBOSE:
        print('[w2l replay] ' + msg, flush=True)


# A torch.cuda.MemPool must not be destroyed while ANY pool is routing allocations (the allocator asserts that no capture is
# under way when it gives a pool's memory back), and Python may collect a dropped record -- or a whole model -- at any moment,
# e.g. inside the next recording.  Every pool is therefore also held here, with a weak reference to the lease object of the
# record set that uses it; purge_pools() -- called at the top of a forward pass, outside every pool context -- lets go of the
# pools whose lease is gone.
_POOLS: list = []


class _Lease:
    __slots__ = ('__weakref__',)


def new_pool(lease):
    pool = torch.cuda.MemPool()
    _POOLS.append((pool, weakref.ref(lease)))
    return pool


def purge_pools():
    if _POOLS and any(ref() is None for _, ref in _POOLS):
        dead = [e for e in _POOLS if e[1]() is None]
        _POOLS[:] = [e for e in _POOLS if e[1]() is not None]
        del dead                                         # (the pools are destroyed here, at a safe point)


class RecordSet:
    """one of the two alternating record sets of a step shape"""

    def __init__(self, group, index):
        self.group = weakref.ref(group)
        self.index = index
        self.reset()

    def reset(self):
        self.lease = _Lease()            # (dropping the old lease releases the old pools at the next purge_pools())
        self.lease_O = _Lease()
        self.lease_X = _Lease()
        self.pool = self.pool_O = self.pool_X = None
        self.F = self.B = self.O = self.X = None
        self.O_sig = self.X_sig = None
        self.x_static = self.x_ref = None
        self.g_static = None
        self.out = None
        self.ectx = None
        self.lens_out = None
        self.lens_static = None          # {'host': pinned int32 [rows, N], 'dev': device copy, 'event': _lib.Event}
        self.grads = None                # [(parameter, static gradient tensor)] in engine.parameters() order (None: no gradient)
        self.deferred = []               # the held-back weight-gradient records of this set's backward pass
        self.O_ptrs = None

    def ensure_pool(self):
        if self.pool is None:
            self.pool = new_pool(self.lease)
        return self.pool


class Group:
    def __init__(self, key):
        self.key = key
        self.sets = [RecordSet(self, 0), RecordSet(self, 1)]
        self.next = 0
        self.seen = 0
        self.failures = 0
        self.disabled = None
        self.fp8_epoch = None
        self.replays = 0
        self.eager_left = 0
        self.last_use = 0


def opt_signature(opt):
    return tuple((g['lr'], g['momentum'], g['weight_decay'], g['nesterov'], g['dampening'], g.get('maximize', False))
                 for g in opt.param_groups)


class StepReplayer:
    """per StackEngine: decides for every training step whether it runs eagerly, is recorded, or is replayed"""

    def __init__(self, engine):
        self.engine = weakref.ref(engine)
        self.groups = {}
        self.pending: Optional[RecordSet] = None      # the set whose backward pass left engine._deferred
        self.counter = None                           # device int64: the dropout step counter of recorded steps
        self.counter_host = 0
        self.clock = 0
        self.after_replayed_O = False                 # an eager phase must first wait for the optimizer's stream
        self.opt_stream = None
        self.first_seen = {}                          # step shapes seen so far that have no group yet -> how often

    # ------------------------------------------------------------------ eligibility
    def _mode_flags(self):
        from . import engine as E
        return (E.DETERMINISTIC_WGRAD, E.DEALT_WGRAD, E.DEFER_SPREAD, E.WGRAD_AFTER_DGRAD, E.FUSED_BN_REDUCE, E.FOLD_BN_FINALIZE,
                E.FOLD_BN_FWD, E.FAST_BN_BWD, E.STAT_SLOTS, E.FP8_DGRAD, E.FP8_WGRAD, E.WGROUP_MAX, E.AUTOTUNE,
                os.environ.get('W2L_WGRAD_GROUPS', 'auto'))

    def plan_forward(self, x, lens, training, softmax_mode, keep_ctx):
        """-> (mode, set) with mode 'replay' / 'record', or None: this forward pass runs eagerly"""
        from . import engine as E
        eng = self.engine()
        if (not ENABLED or eng is None or not training or keep_ctx or not x.is_cuda or x.requires_grad or not torch.is_grad_enabled()
                or eng.head is None or eng.precise or eng.grad_ready is not None or eng.flat_ready is not None
                or eng.backward_done is not None or eng.grad_reduce_start is not None or eng.dropout_counter is not None
                or E.KERNEL_TIMER is not None or E.JOIN_EVENTS is not None or _lib._trace['saved'] or _lib.recording() is not None
                or x.dtype != torch.float32 or not x.is_contiguous() or (lens is not None and lens.is_cuda)
                or E.DEFER_SPREAD != 'start' or torch.cuda.is_current_stream_capturing()):
            return None
        purge_pools()
        defer = eng.defer_wgrad if eng.deferred is not None else 0
        key = (tuple(x.shape), softmax_mode, lens is None, x.device.index, _lib.raw_stream(), self._mode_flags(),
               defer if isinstance(defer, int) else tuple(sorted(defer)), id(eng.deferred), bool(eng.overlap_wgrad),
               torch.initial_seed(), E._wev_epoch[0])
        self.clock += 1
        g = self.groups.get(key)
        if g is None:
            # a shape gets a group (and with it, once warm, two sets of static buffers) only after it has COME BACK: with
            # variable-length batches most shapes are seen once, and must neither cost a recording nor push a hot shape out
            n = self.first_seen.get(key, 0) + 1
            if len(self.first_seen) > 4096:
                self.first_seen.clear()
            self.first_seen[key] = n
            if n <= WARM_STEPS:
                return None
            if len(self.groups) >= MAX_GROUPS:
                victim = min(self.groups.values(), key=lambda v: v.last_use)
                if (self.pending is not None and self.pending.group() is victim) or self.clock - victim.last_use < 4 * MAX_GROUPS:
                    return None                        # (every recorded shape is in recent use: this one stays eager)
                del self.groups[victim.key]
            free, total = torch.cuda.mem_get_info(x.device)
            if free < 0.2 * total:                     # static buffers double a step's activations: not on a nearly full device
                return None
            g = self.groups[key] = Group(key)
            g.seen = WARM_STEPS
            g.fp8_epoch = E._fp8_epoch[0]
        g.last_use = self.clock
        if g.disabled is not None:
            return None
        # (kernel plans: a step that measured any -- its own first steps -- is not recorded, _record_forward / _record_backward
        # check the plan tables around the step; plans measured for OTHER shapes later on change nothing a record names, so a
        # growing table must not -- and does not -- invalidate the shapes already recorded: variable-length training keeps
        # meeting new shapes)
        if g.fp8_epoch != E._fp8_epoch[0]:             # an e4m3 weight scale moved: recorded scales (by value) are stale
            g.fp8_epoch = E._fp8_epoch[0]
            self._drop(g)
        g.seen += 1
        if g.seen <= WARM_STEPS:
            return None
        if eng.fp8:
            if g.eager_left > 0:
                g.eager_left -= 1
                return None
            if g.replays >= FP8_EAGER_EVERY:
                g.replays = 0
                g.eager_left = FP8_EAGER_STEPS - 1
                self._age_fp8_weights(FP8_EAGER_EVERY)
                return None
        s = g.sets[g.next]
        if self.pending is s:                          # (never run a set whose own held-back gradients are still pending)
            return None
        g.next ^= 1
        return ('replay' if s.F is not None and s.B is not None else 'record'), s

    def _age_fp8_weights(self, n):
        eng = self.engine()
        for u in eng.units:
            for c in (u.main, u.res):
                st = c.weight.__dict__.get('_w2l_fp8') if c is not None else None
                if st is not None:
                    st['age'] += n

    def _drop(self, g):
        for s in g.sets:
            if self.pending is s:
                continue                               # its deferred records are live Python objects: they stay valid
            s.reset()

    def fail(self, s: RecordSet, why: str):
        g = s.group()
        s.reset()
        if g is None:
            return
        STATS['poisoned'].append(why)
        _say(f'recording dropped: {why}')
        if any(t in why for t in TRANSIENT):
            g.seen = 0                                 # something that passes by itself: warm steps again, then another try
            return
        g.failures += 1
        if g.failures >= MAX_FAILURES:
            g.disabled = why
            for t in g.sets:
                if self.pending is not t:
                    t.reset()

    # ------------------------------------------------------------------ eager neighbours
    def before_eager(self):
        """an eager phase follows replayed ones: the per-weight 'update pending' flags were not kept while replaying, so the
        caller's stream waits for the optimizer's stream as a whole"""
        if self.after_replayed_O and self.opt_stream is not None:
            _lib.stream_wait_stream(_lib.raw_stream(), self.opt_stream)
        self.after_replayed_O = False

    # ------------------------------------------------------------------ phase X
    def flush_pending(self):
        """the held-back weight gradients of the last backward pass, in front of a forward pass: replayed, recorded, or left to
        the eager engine (engine.forward calls flush_deferred itself)"""
        eng = self.engine()
        s = self.pending
        if s is None or not eng._deferred:
            self.pending = None
            return
        opt = eng.deferred
        if opt is None or not all(opt.stepped(r['token']) for r in eng._deferred) or s.B is None:
            self.pending = None
            return                                     # eager flush_deferred sorts it out (stale gradients, no optimizer)
        # what the held-back updates are applied with: the hyper-parameters of the LAST step() call (optim.FusedSGD.apply), by
        # value in the record -- not necessarily the optimizer's current ones (a scheduler may have stepped since)
        hp = opt._deferred_state()['hp']
        sig = tuple(hp[id(r['conv'].weight)][2] for r in eng._deferred)
        if s.X is not None and s.X_sig != sig:
            s.X = None
        if s.X is not None:
            s.X.replay()
            eng._deferred = []
            eng._side_used = True
            STATS['replayed_X'] += 1
            self.pending = None
            return
        s.lease_X = _Lease()
        s.pool_X = new_pool(s.lease_X)                 # (a pool of its own: see the module docstring)
        with torch.cuda.use_mem_pool(s.pool_X), _lib.Recorder() as rec:
            eng.flush_deferred(pos=0)
        ph = rec.finish()
        self.pending = None
        if ph is None or eng._deferred:
            s.pool_X = None
            g = s.group()
            if g is not None:
                g.failures += 1
                if g.failures >= MAX_FAILURES:
                    g.disabled = rec.poisoned or 'held-back weight gradients not launched in one piece'
            STATS['poisoned'].append(rec.poisoned or 'phase X incomplete')
            return
        s.X, s.X_sig = ph, sig

    # ------------------------------------------------------------------ dropout counter
    def sync_counter(self, dev, advance):
        """the device step counter follows the host's count of dropout draws (engine._dropout_calls), so eager and replayed
        steps never reuse an offset; returns the counter tensor"""
        from . import engine as E
        if self.counter is None or self.counter.device != dev:
            self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
            self.counter_host = 0
        if self.counter_host != E._dropout_calls:
            self.counter.fill_(E._dropout_calls)
            self.counter_host = E._dropout_calls
        E._dropout_calls += advance
        self.counter_host += advance
        return self.counter


def replayer_for(engine) -> Optional[StepReplayer]:
    if not ENABLED:
        return None
    rp = engine.__dict__.get('_replayer')
    if rp is None:
        rp = engine.__dict__['_replayer'] = StepReplayer(engine)
    return rp


# ---------------------------------------------------------------------------------------------------------------- lengths
def lens_rows_host(engine, lens):
    """Jasper's length chain (jasper.py:109-121: lens <- (lens + 2p - d(k-1) - 1) / s + 1 in float, truncated where a mask is
    applied) for host lengths, in numpy float32 with the same operation order as engine._plan_lens: (int32 rows [R, N] in the
    order first, per-unit depthwise lengths, per-unit output lengths; which units have which; the final float lengths)"""
    cur = lens.numpy().astype(np.int32)
    cur_f = cur.astype(np.float32)
    rows, mid_has, out_has = [cur], [], []
    one = np.float32(1.0)
    mids = []
    outs = []
    for u in engine.units:
        has_mid = False
        if u.dw is not None and u.update_lens:
            c = u.dw
            cur_f = (cur_f + np.float32(c.pad_l + c.pad_r) - np.float32(c.dilation * (c.kernel - 1)) - one) / np.float32(c.stride) + one
            cur = cur_f.astype(np.int32)
            mids.append(cur)
            has_mid = True
        mid_has.append(has_mid)
        if u.update_lens:
            c = u.main
            cur_f = (cur_f + np.float32(c.pad_l + c.pad_r) - np.float32(c.dilation * (c.kernel - 1)) - one) / np.float32(c.stride) + one
            cur = cur_f.astype(np.int32)
        out_has.append(bool(u.mask_out))
        if u.mask_out:
            outs.append(cur)
    return np.stack(rows + mids + outs), mid_has, out_has, torch.from_numpy(cur_f.copy())


# ---------------------------------------------------------------------------------------------------------------- autograd
class _ReplayFn(torch.autograd.Function):
    """the whole conv stack + classifier + (log_)softmax as one autograd node whose forward and backward are recorded launch
    lists (or are being recorded right now)"""

    @staticmethod
    def forward(ctx, x, engine, lens, softmax_mode, holder, mode, rset, *params):
        rp = replayer_for(engine)
        ctx.engine, ctx.rset, ctx.mode, ctx.softmax_mode = engine, rset, mode, softmax_mode
        if mode == 'replay':
            out = _replay_forward(rp, engine, rset, x, lens)
        else:
            out = _record_forward(rp, engine, rset, x, lens, softmax_mode)     # (a dropped recording still ran the step in full)
        holder['lens_out'] = rset.lens_out
        return out

    @staticmethod
    def backward(ctx, g):
        engine, rset = ctx.engine, ctx.rset
        rp = replayer_for(engine)
        n = len(engine.parameters())
        if any(p.grad is not None for p, _ in (rset.grads or [])) or (rset.grads is None and
                                                                      any(p.grad is not None for p in engine.parameters())):
            # gradient accumulation: autograd must ADD -- the eager backward on this set's saved context does that
            if rset.ectx is None:
                raise RuntimeError('replayed step: p.grad is already set (gradient accumulation) and the eager context of this '
                                   'record set is gone; call optimizer.zero_grad(set_to_none=True) before backward(), or set W2L_REPLAY=0')
            rp.before_eager()
            grads = engine.backward(rset.ectx, g)
            rp.pending = None
            _last_backward[0] = None
            return (None,) * 7 + tuple(grads)
        if ctx.mode == 'replay':
            _replay_backward(rp, engine, rset, g)
        else:
            _record_backward(rp, engine, rset, g)
        return (None,) * (7 + n)


def _upload_lens(rset, engine, lens):
    rows, mid_has, out_has, final = lens_rows_host(engine, lens)
    st = rset.lens_static
    if st is None or tuple(st['host'].shape) != rows.shape:
        return None, None
    # a ring of pinned staging buffers: the copy issued from a buffer LENS_RING uses of this set ago (2 x LENS_RING steps) is
    # long done, so the wait below never holds the host back
    ring = st.setdefault('ring', [(st['host'], st['event'])])
    if len(ring) < LENS_RING:
        ring.append((torch.empty(rows.shape, dtype=torch.int32, pin_memory=True), _lib.Event()))
        host, ev = ring[-1]
    else:
        st['turn'] = (st.get('turn', 0) + 1) % LENS_RING
        host, ev = ring[st['turn']]
        ev.synchronize()
    host.numpy()[...] = rows
    st['dev'].copy_(host, non_blocking=True)
    ev.record()
    return st, final


def _replay_forward(rp, engine, rset, x, lens):
    from . import engine as E
    if lens is not None:
        st, final = _upload_lens(rset, engine, lens)
        if st is None:
            raise RuntimeError('replayed step: the length table of this record set does not fit the batch')
        rset.lens_out = final
    rset.x_static.copy_(x, non_blocking=True)
    rp.flush_pending()
    if engine._deferred:                               # what an eager backward held back: launched eagerly, as engine.forward would
        rp.before_eager()
        engine.flush_deferred(pos=0)
    rp.sync_counter(x.device, len(engine.units) + 1)
    rset.F.replay()
    g = rset.group()
    if g is not None:
        g.replays += 1
    STATS['replayed_F'] += 1
    return rset.out.detach()


def _record_forward(rp, engine, rset, x, lens, softmax_mode):
    from . import engine as E
    rp.before_eager()
    rp.flush_pending()
    if engine._deferred:                               # something an eager backward left: launched eagerly, outside the record
        engine.flush_deferred(pos=0)
    rset.reset()
    pool = rset.ensure_pool()
    n_drop = len(engine.units) + 1
    counter = rp.sync_counter(x.device, n_drop)
    tune0 = (len(E._tuned_shapes), len(E._wgroup_forms))
    with torch.cuda.use_mem_pool(pool):
        rset.x_static = torch.empty_like(x)
        rset.x_static.copy_(x)
        rset.x_ref = None
        if lens is not None:
            rows, _, _, _ = lens_rows_host(engine, lens)
            rset.lens_static = {'host': torch.empty(rows.shape, dtype=torch.int32, pin_memory=True),
                                'dev': torch.empty(rows.shape, dtype=torch.int32, device=x.device), 'event': _lib.Event()}
        engine.dropout_counter = counter
        engine._lens_static = rset.lens_static
        try:
            with _lib.Recorder() as rec:
                out, ectx = engine.forward(rset.x_static, lens, True, softmax_mode, want_input_grad=False)
                # the step drew its masks at offsets counter + unit index: the NEXT step -- replayed or eager -- starts
                # behind them (the host's count of draws, engine._dropout_calls, was advanced by the same amount)
                check(lib.w2l_counter_add(ptr(counter), n_drop, _lib.raw_stream()), 'w2l_counter_add')
        finally:
            engine.dropout_counter = None
            engine._lens_static = None
    ph = rec.finish()
    rset.ectx, rset.out, rset.lens_out = ectx, out, ectx['lens_out']
    if ph is not None and tune0 != (len(E._tuned_shapes), len(E._wgroup_forms)):
        ph, rec.poisoned = None, 'kernel plans were measured during the step'
    if ph is None:
        why = rec.poisoned
        # this step still completes eagerly: its context is whole, only the list is dropped
        rset_out, rset_ctx = out, ectx
        rp.fail(rset, 'forward: ' + str(why))
        rset.ectx, rset.out, rset.lens_out = rset_ctx, rset_out, rset_ctx['lens_out']
        return out
    rset.F = ph
    return out


def _assign_grads(rset):
    for p, gbuf in rset.grads:
        if gbuf is not None:
            p.grad = gbuf.detach()


def _record_backward(rp, engine, rset, g):
    from . import engine as E
    pool = rset.ensure_pool()
    tune0 = (len(E._tuned_shapes), len(E._wgroup_forms))
    with torch.cuda.use_mem_pool(pool):
        rset.g_static = torch.empty(g.shape, dtype=torch.float32, device=g.device)
        rset.g_static.copy_(g)
        with _lib.Recorder() as rec:
            grads = engine.backward(rset.ectx, rset.g_static)
    ph = rec.finish() if rset.F is not None else None
    params = engine.parameters()
    why = rec.poisoned
    if ph is not None and tune0 != (len(E._tuned_shapes), len(E._wgroup_forms)):
        ph, why = None, 'kernel plans were measured during the step'
    fixed = []
    for p, gr in zip(params, grads):
        # autograd stores a gradient whose strides are not the parameter's (the sliced dW of a padded channel count) as a copy
        # in the parameter's layout -- and optim.FusedSGD picks its update path by that layout: do the same here, so that a
        # step that was being recorded leaves exactly the eager step's state (such a step is not replayable: a torch copy)
        if gr is not None and (gr.shape != p.shape or any(a != b for n, a, b in zip(p.shape, gr.stride(), p.stride()) if n > 1)):
            gr = torch.empty_like(p).copy_(gr)
            ph, why = None, 'gradient of a padded channel count'
        fixed.append((p, gr))
    rset.grads = fixed
    _assign_grads(rset)
    rset.deferred = list(engine._deferred)
    if ph is None:
        if rset.F is not None:
            rp.fail(rset, 'backward: ' + str(why))
        # (the eager backward has run in full: gradients are assigned, held-back records are the engine's)
        rp.pending = None
        _last_backward[0] = None
        return
    rset.B = ph
    rp.pending = rset if rset.deferred else None
    _last_backward[0] = (weakref.ref(rp), rset)
    STATS['recorded'] += 1
    _say(f'recorded set {rset.index}: F {rset.F.n_calls} calls, B {rset.B.n_calls} calls, {len(rset.deferred)} held-back gradients')


def _replay_backward(rp, engine, rset, g):
    rset.g_static.copy_(g, non_blocking=True)
    rset.B.replay()
    engine._side_used = False
    if rset.deferred:
        opt = engine.deferred
        tok = opt.token() if opt is not None else 0
        for r in rset.deferred:
            r['token'] = tok
        engine._deferred = list(rset.deferred)
        rp.pending = rset
    else:
        rp.pending = None
    _assign_grads(rset)
    _last_backward[0] = (weakref.ref(rp), rset)
    STATS['replayed_B'] += 1


# ---------------------------------------------------------------------------------------------------------------- optimizer
def optimizer_step(opt, eager_body) -> bool:
    """optim.FusedSGD.step(): replay the recorded phase O of the set whose backward pass just ran, or record it while the eager
    body runs.  Returns True if the step was taken here."""
    hit = _last_backward[0]
    _last_backward[0] = None
    if hit is None or not ENABLED:
        return False
    rp, rset = hit[0](), hit[1]
    if rp is None or rset.B is None or rset.grads is None:
        return False
    engine = rp.engine()
    if engine is None:
        return False
    mine = opt.__dict__.get('_w2l_param_ids')
    if mine is None:
        mine = opt.__dict__['_w2l_param_ids'] = frozenset(id(p) for g in opt.param_groups for p in g['params'])
    if any(id(p) not in mine for p, _ in rset.grads):
        return False
    sig = (opt_signature(opt), bool(opt.overlap), _lib.raw_stream())
    if rset.O is not None and rset.O_sig == sig:
        for (p, gbuf), want in zip(rset.grads, rset.O_ptrs):
            have = p.grad.data_ptr() if p.grad is not None else 0
            if have != want:
                return False                           # someone replaced a gradient tensor: the eager step takes what is there
        rset.O.replay()
        st, dst = opt._side_state(), opt._deferred_state()
        dst['seq'] += 1
        st['pending'] = True
        rp.after_replayed_O = True
        rp.opt_stream = st['stream']
        STATS['replayed_O'] += 1
        return True
    # record: the eager body under the recorder, allocating from a pool of the phase's own
    rset.lease_O = _Lease()
    rset.pool_O = new_pool(rset.lease_O)
    with torch.cuda.use_mem_pool(rset.pool_O), _lib.Recorder() as rec:
        eager_body()
    ph = rec.finish()
    if ph is None:
        rset.pool_O = None
        g = rset.group()
        if g is not None:
            g.failures += 1
            if g.failures >= MAX_FAILURES:
                g.disabled = 'optimizer step: ' + str(rec.poisoned)
        STATS['poisoned'].append('optimizer step: ' + str(rec.poisoned))
        _say('optimizer phase dropped: ' + str(rec.poisoned))
        return True
    rset.O, rset.O_sig = ph, sig
    rset.O_ptrs = [p.grad.data_ptr() if p.grad is not None else 0 for p, _ in rset.grads]
    rp.opt_stream = opt._side_state()['stream']
    return True


def report(engine) -> dict:
    """what the replayer of ``engine`` holds (diagnostics; bench.py puts it into the record)"""
    rp = engine.__dict__.get('_replayer')
    out = {'enabled': ENABLED, 'shapes': []}
    if rp is None:
        return out
    for g in rp.groups.values():
        out['shapes'].append({'input': list(g.key[0]), 'seen': g.seen, 'disabled': g.disabled, 'failures': g.failures,
                              'sets': [{'F': s.F.n_calls if s.F else None, 'B': s.B.n_calls if s.B else None,
                                        'O': s.O.n_calls if s.O else None, 'X': s.X.n_calls if s.X else None,
                                        'python_items': sum(1 for ph in (s.F, s.B, s.O, s.X) if ph for it in ph.items if it[0] == 'py')}
                                       for s in g.sets]})
    return out
