"""FusedSGD: torch.optim.SGD semantics (the reference's optimizer, configuration/optimizer/
exp_lr_optimizer.yaml:2-7) with the conv-weight update fused with the bf16 operand packing of the
next step (w2l_sgd_pack).  Non-conv parameters (biases, BatchNorm affine) use torch's own foreach
update.  State-dict layout is torch.optim.SGD's (``momentum_buffer`` per parameter).

The conv-weight updates are HBM-bound (24 B per parameter, 3.7 GB per step for the full Wav2Letter table) while the
forward convolutions that follow them are MFMA-bound, so ``step()`` enqueues them on a side HIP stream in forward order
and tags each weight's operand pack with an event; the step engine waits for a layer's event just before that layer's
first convolution (engine.pack_weights).  The next forward therefore starts as soon as layer 0 is updated, and the other
20 updates stream through HBM underneath it.  This is opt-in (``optimizer.overlap = True``; trainer.Trainer and bench.py
do): ``join()`` (also called by ``state_dict``) makes the caller's stream wait for the updates in flight, and anything
that reads the parameters outside the step engine must call it first.

``defer_wgrad(model, k)`` goes one step further: the weight gradients of the model's top ``k`` conv units are not computed
in backward at all but at the start of the NEXT forward pass, on the weight-gradient stream, each followed by this
optimizer's fused update of that weight (engine.StackEngine.flush_deferred).  The top layers are the first of the backward
pass and the last of the forward pass, so their gradients are the ones with slack; launched beside the forward they give
the matrix cores work while the forward's own BatchNorm / CTC kernels run.  One optimizer step per batch, as before: a
layer's forward convolution waits for the event behind its update.  ``p.grad`` of those weights stays ``None`` (the update
consumes the gradient directly); ``join()`` flushes whatever is pending."""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib
from . import engine as E
from ._lib import check, lib, ptr, stream_ptr


def _is_tap_major(t: torch.Tensor) -> bool:
    """logical [Cout, Cin, Kw] view of a dense [Kw, Cout, Cin] storage"""
    if t.dim() != 3:
        return False
    co, ci, kw = t.shape
    return t.stride() == (ci, 1, co * ci)


class FusedSGD(torch.optim.SGD):
    # W2L_RECYCLE_GRADS=1: zero the consumed conv-weight gradients in the update kernel and hand them back as the next dW
    # (no fill launches for split-K weight gradients).  Off by default: measured NEUTRAL on the full Wav2Letter table
    # (13.54 vs 13.50 ms per step) -- the fills already ran hidden on the weight-gradient stream.
    recycle_grads = os.environ.get('W2L_RECYCLE_GRADS', '0') == '1'
    overlap = False           # opt-in (trainer.Trainer and bench.py set it): whoever enables it must join() before
                              # reading parameters outside the step engine (checkpoints, .cpu() copies, ...)

    @classmethod
    def from_sgd(cls, opt: torch.optim.SGD) -> 'FusedSGD':
        new = cls.__new__(cls)
        new.__dict__.update(opt.__dict__)
        return new

    def _side_state(self):
        st = self.__dict__.get('_w2l_side')
        if st is None:
            st = {'stream': None, 'held': [], 'pending': False, 'packs': []}
            self.__dict__['_w2l_side'] = st
        return st

    # ------------------------------------------------------------------ deferred weight gradients
    def _deferred_state(self):
        st = self.__dict__.get('_w2l_deferred')
        if st is None:
            st = {'seq': 0, 'hp': {}, 'engines': []}
            self.__dict__['_w2l_deferred'] = st
        return st

    def defer_wgrad(self, model, units):
        """hold back the weight gradients of some of ``model``'s conv units until its next forward pass: an int k = the top k
        units, an iterable of unit indices (negative: counted from the top) = exactly those; 0 / empty switches it off"""
        self.join()
        units = int(units) if isinstance(units, int) else frozenset(int(u) for u in units)
        model._defer_wgrad = units
        model._deferred_opt = self if units else None

    def _register_engine(self, eng):
        # weak references: a model that rebuilds its engine (module.to(), load_state_dict of replaced tensors) must not keep
        # every dead engine -- its streams and held tensors -- alive through the optimizer
        st = self._deferred_state()
        st['engines'] = [r for r in st['engines'] if r() is not None]
        if not any(r() is eng for r in st['engines']):
            st['engines'].append(weakref.ref(eng))

    def _engines(self):
        return [e for e in (r() for r in self._deferred_state()['engines']) if e is not None]

    def zero_grad(self, set_to_none: bool = True):
        """torch's zero_grad, plus: weight gradients held back by a backward pass that no step() followed are dropped with
        the rest (a skipped step -- non-finite loss guard, manual skip -- must not leak into the next one)"""
        for eng in self._engines():
            eng.drop_unstepped()
        return super().zero_grad(set_to_none=set_to_none)

    def accepts(self, p) -> bool:
        """would step() take this parameter through the fused conv-weight update?"""
        hp = self._deferred_state()['hp'].get(id(p))
        if hp is None:
            for group in self.param_groups:
                ok = group['momentum'] != 0 and group['dampening'] == 0 and not group.get('maximize', False)
                for q in group['params']:
                    self._deferred_state()['hp'][id(q)] = [ok, group, None]
            hp = self._deferred_state()['hp'].get(id(p))
        return bool(hp is not None and hp[0] and self.overlap and p.is_cuda and p.dtype == torch.float32 and _is_tap_major(p)
                    and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0)

    def token(self) -> int:
        return self._deferred_state()['seq']

    def stepped(self, token: int) -> bool:
        """has step() run since ``token`` was drawn (i.e. since the backward pass that deferred a gradient)?"""
        return self._deferred_state()['seq'] > token

    @torch.no_grad()
    def apply(self, p, g):
        """the fused update of ONE conv weight with gradient ``g``, on the current stream (the stream that produced ``g``),
        with the hyper-parameters of the last step() call"""
        ok, group, hp = self._deferred_state()['hp'][id(p)]
        lr, mu, wd, nesterov = hp
        pk = self._fused_conv(p, g, lr, mu, wd, nesterov)
        pk.ready = E.weight_event(p)
        pk.ready.record()

    def join(self):
        """make the current stream wait for every update of the last step: deferred weight gradients are launched and
        applied first, then the weight-gradient stream and the optimizer's side stream are joined"""
        for eng in self._engines():
            eng.flush_deferred()
            eng.join_side()
        self._join_updates()

    def _join_updates(self):
        """make the current stream wait for the updates still running on the optimizer's side stream"""
        st = self._side_state()
        if st['pending'] and st['stream'] is not None:
            _lib.stream_wait_stream(_lib.raw_stream(), st['stream'])
        st['pending'] = False
        st['held'], st['packs'] = [], []
        if self._release_held in E.AFTER_FORWARD:
            E.AFTER_FORWARD.remove(self._release_held)

    def _release_held(self):
        """engine hook, end of a forward pass: once that forward has waited for every update event of the last step() the
        gradients those updates read (0.6 GB for the full Wav2Letter table) can go back to the allocator"""
        st = self._side_state()
        if all(pk.ready is None for pk in st['packs']):
            st['held'], st['packs'] = [], []
            if self._release_held in E.AFTER_FORWARD:
                E.AFTER_FORWARD.remove(self._release_held)

    def state_dict(self):
        self.join()
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # the gradients of a recorded / replayed backward pass: the step is replayed as ONE w2l_replay call, or recorded now
        from . import replay
        if replay._last_backward[0] is not None and replay.optimizer_step(self, self._step_eager):
            return loss
        self._step_eager()
        return loss

    def _step_eager(self):
        rp_hit = None
        for eng in self._engines():
            rp_hit = eng.__dict__.get('_replayer') or rp_hit
        if rp_hit is not None:
            rp_hit.before_eager()
        self._join_updates()         # (normally a no-op: the forward pass has already waited for every event)
        for eng in self._engines():  # a weight with BOTH a held-back gradient and a .grad gets one update with their sum
            eng.settle_before_step()
        st = self._side_state()
        dst = self._deferred_state()
        dst['seq'] += 1              # gradients deferred by the backward pass just run now count as "stepped"
        for group in self.param_groups:
            lr, mu, wd = group['lr'], group['momentum'], group['weight_decay']
            nesterov, dampening, maximize = group['nesterov'], group['dampening'], group.get('maximize', False)
            for p in group['params']:                     # what a deferred update of this step is applied with
                ent = dst['hp'].get(id(p))
                if ent is not None:
                    ent[2] = (lr, mu, wd, nesterov)
            fused_ok = mu != 0 and dampening == 0 and not maximize
            rest, fused = [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad
                if (fused_ok and p.is_cuda and p.dtype == torch.float32 and _is_tap_major(p) and g.stride() == p.stride()
                        and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0):
                    fused.append((p, g))
                else:
                    rest.append(p)
            if rest:
                self._plain(rest, lr, mu, wd, nesterov, dampening, maximize)
            if not fused:
                continue
            if not self.overlap:
                for p, g in fused:
                    self._fused_conv(p, g, lr, mu, wd, nesterov)
                continue
            dev = fused[0][0].device
            if st['stream'] is None or st['stream'].device != dev:
                from .streams import concurrent_stream        # measured to run beside the main and weight-gradient streams
                _lib.poison('optimizer side stream created')
                st['stream'] = concurrent_stream(dev, 'sgd')
            side = st['stream']
            _lib.stream_wait_stream(side, _lib.raw_stream())     # gradients (wgrad join, all-reduce) are complete there
            with torch.cuda.stream(side):
                for p, g in fused:               # parameter order = forward order: layer 0's event fires first
                    pk = self._fused_conv(p, g, lr, mu, wd, nesterov)
                    pk.ready = E.weight_event(p)
                    pk.ready.record(side)
                    st['held'].append(g)         # zero_grad() must not hand this memory back while the kernel reads it
                    st['packs'].append(pk)
            st['pending'] = True
            if self._release_held not in E.AFTER_FORWARD:
                E.AFTER_FORWARD.append(self._release_held)

    def _fused_conv(self, p, g, lr, mu, wd, nesterov):
        state = self.state[p]
        first = 'momentum_buffer' not in state or state['momentum_buffer'] is None
        if first:
            _lib.poison('momentum buffer created')
            state['momentum_buffer'] = torch.empty_like(p)           # preserves the tap-major strides
        buf = state['momentum_buffer']
        if buf.stride() != p.stride():
            _lib.poison('momentum buffer re-laid out')
            buf = torch.empty_like(p).copy_(buf)
            state['momentum_buffer'] = buf
        cout, cin, kw = p.shape
        dev = p.device
        cache = getattr(p, '_w2l_pack', None)
        if cache is None:
            cache = E._Volatile()
            p._w2l_pack = cache
        precise = any(k for k in cache)                              # keep hi/lo pairs alive only if fp32 mode is in use
        old = cache.get(precise)
        if old is not None and old.fwd_hi.device == dev and old.coutp == cout and old.cinp == cin:
            fwd_hi, fwd_lo, dgr_hi, dgr_lo = old.fwd_hi, old.fwd_lo, old.dgr_hi, old.dgr_lo
        else:
            _lib.poison('operand pack buffers created')
            fwd_hi = torch.empty(kw, cout, cin, dtype=torch.bfloat16, device=dev)
            dgr_hi = torch.empty(kw, cin, cout, dtype=torch.bfloat16, device=dev)
            fwd_lo = torch.empty_like(fwd_hi) if precise else None
            dgr_lo = torch.empty_like(dgr_hi) if precise else None
        # the gradient buffer is handed back to the step engine zero-filled (engine._wgrad_now takes it as the next dW when
        # zero_grad(set_to_none=True) has dropped p.grad): no fill launch for the split-K weight gradients of the next step
        recycle = self.recycle_grads and not precise
        # fp8 mode (engine._fp8_weights left its state on the parameter): emit the e4m3 operands of the next step here too,
        # with the scale in force -- its periodic re-derivation from amax stays with the engine (asynchronous: a new scale
        # is adopted at a forward pass, which then requantises both layouts itself for that one step)
        f8 = p.__dict__.get('_w2l_fp8')
        if f8 is not None and (precise or f8['q'].shape != fwd_hi.shape or f8['q'].device != dev):
            f8 = None
        check(lib.w2l_sgd_pack(ptr(p), ptr(g), ptr(buf), int(first), float(lr), float(mu), float(wd), int(nesterov),
                               int(recycle), cout, cin, kw, ptr(fwd_hi), ptr(fwd_lo), ptr(dgr_hi), ptr(dgr_lo),
                               ptr(f8['q']) if f8 else None, ptr(f8['qd']) if f8 else None, f8['scale'] if f8 else 1.0,
                               stream_ptr()), 'w2l_sgd_pack')
        if recycle:
            p._w2l_dw_zeroed = g.permute(2, 0, 1)                    # the dense [Kw, Cout, Cin] storage of g
        torch.autograd.graph.increment_version(p)                    # p changed through its raw pointer
        cache.clear()
        pk = E._PackedW(p._version, fwd_hi, fwd_lo, dgr_hi, dgr_lo, cin, cout, p.data_ptr())
        cache[precise] = pk
        if f8 is not None:                      # both e4m3 layouts are current for this version
            f8['version'] = f8['version_d'] = pk.version
            f8['age'] += 1
        return pk

    def _small_multi(self, params, lr, mu, wd, nesterov) -> bool:
        """torch.optim.SGD's update of all the small parameters (conv biases, BatchNorm gamma / beta) in ONE launch
        (w2l_sgd_small_multi) instead of five torch._foreach_* calls over ~60 tensors: a device table of (p, g, m, n) built once
        per set of addresses (``params``: those _multi_ok admits); an entry point, so a recorded launch list replays it."""
        rows = []
        for p in params:
            g = p.grad
            m = self.state[p].get('momentum_buffer') if mu != 0 else None
            rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else 0, p.numel()))
        key = tuple(rows)
        cache = self.__dict__.setdefault('_w2l_small_tables', {})
        table = cache.get(key)
        if table is None:
            # (built while the optimizer phase of a step is being recorded, the table lives in that phase's own memory pool:
            # a live allocation nothing else of the record ever writes -- no reason to drop the recording)
            if len(cache) > 8:
                cache.clear()
            flat = []
            for pp, gp, mp, n in rows:
                flat += [pp, gp, mp, n]                    # w2l_sgd_small_t: three pointers, then {int32 n, int32 pad} = one int64 < 2^31
            table = cache[key] = torch.tensor(flat, dtype=torch.int64).to(params[0].device)
        check(lib.w2l_sgd_small_multi(ptr(table), len(rows), max(r[3] for r in rows), float(lr), float(mu), float(wd), int(nesterov),
                                      stream_ptr()), 'w2l_sgd_small_multi')
        for p in params:
            torch.autograd.graph.increment_version(p)
        return True

    def _multi_ok(self, p, mu) -> bool:
        """may this parameter take the one-launch elementwise update?  fp32 on the device, dense, and gradient / momentum
        buffer in the parameter's own physical layout (contiguous, or -- a depthwise conv weight -- the tap-major permutation)"""
        def same_layout(a, b):          # (the stride of a size-1 dimension means nothing)
            return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)

        g = p.grad
        if not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.device != p.device or not same_layout(g, p):
            return False
        if not (p.is_contiguous() or (p.dim() == 3 and p.permute(2, 0, 1).is_contiguous())):
            return False
        if mu != 0:
            m = self.state[p].get('momentum_buffer')
            if m is None or not same_layout(m, p) or m.dtype != torch.float32:
                return False
        return True

    def _plain(self, params, lr, mu, wd, nesterov, dampening, maximize):
        # (not under hipGraph capture: the gradient addresses -- and with them the table -- are the capture's own, and building
        # the table is a host-to-device copy from pageable memory, which a capturing stream refuses)
        if not maximize and dampening == 0 and params and not torch.cuda.is_current_stream_capturing():
            multi = [p for p in params if self._multi_ok(p, mu)]
            if multi and self._small_multi(multi, lr, mu, wd, nesterov):
                if len(multi) == len(params):
                    return
                taken = set(id(p) for p in multi)
                params = [p for p in params if id(p) not in taken]
        _lib.poison('torch foreach update of the small parameters')
        grads = [p.grad for p in params]
        if maximize:
            grads = torch._foreach_neg(grads)
        if wd != 0:
            grads = torch._foreach_add(grads, params, alpha=wd)
        if mu != 0:
            bufs = []
            fresh = []
            for p, g in zip(params, grads):
                st = self.state[p]
                if st.get('momentum_buffer') is None:
                    st['momentum_buffer'] = torch.clone(g).detach()
                    fresh.append(True)
                else:
                    fresh.append(False)
                bufs.append(st['momentum_buffer'])
            old = [b for b, f in zip(bufs, fresh) if not f]
            oldg = [g for g, f in zip(grads, fresh) if not f]
            if old:
                torch._foreach_mul_(old, mu)
                torch._foreach_add_(old, oldg, alpha=1 - dampening)
            if nesterov:
                grads = torch._foreach_add(grads, bufs, alpha=mu)
            else:
                grads = bufs
        torch._foreach_add_(params, grads, alpha=-lr)
